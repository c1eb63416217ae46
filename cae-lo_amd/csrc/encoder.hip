// encoder.hip -- 3D-CAE voxel-patch descriptor encoder on the gfx950 matrix cores (f32 in / out / accumulate).
//
// Reference behaviour restated here (never its code): PatchEncoder.predict via
// GetFeaturesFromPatches (Match.py:130-135) with the shipped EncoderModel4VoxelPatch.h5:
//   Conv3D(1->8,3^3,same) tanh -> MaxPool3D(2) -> Conv3D(8->16) tanh -> MaxPool3D(2)
//   -> Conv3D(16->32) tanh -> Flatten(x,y,z,c) -> Dense(200) tanh -> Dense(20) tanh
// (Keras channels-last, zero 'same' padding, cross-correlation; the stale relu/linear script
// AE4VoxelPatch.py:177-197 is NOT what ships).  7.905 MFLOP per patch, 24.28 GFLOP
// per 3072-patch frame: MFMA-bound (157 TFLOP/s f32 matrix peak; conv3 and Dense(200) go through the 16x faster bf16
// pipe with exact 3-way operand splits, see k_enc_conv3).
//
// Kernels
//   k_enc_stage1  persistent workgroups, one patch at a time from a global work counter (coarsest scale first):
//                 bit-packed patch -> conv1+pool1 evaluated only where the 4^3 receptive field holds a set voxel
//                 (binary input: a sum of weight rows; masks built by scattering the set voxels) -> the difference
//                 to the background response in LDS (two 4-channel planes, halo) -> conv2 as implicit GEMM on
//                 v_mfma_f32_16x16x4_f32, all 54 B-fragments of W2 resident in VGPRs, all-zero input rows skipped
//                 (23 % of the dense MFMAs run) -> pool2 in registers (tanh(max) == max(tanh)) ->
//                 P2 [patch][4][4][4][16] to HBM (4 KB).
//   k_enc_conv3   conv3 implicit GEMM, 2 patches per workgroup; f32 products as 6 bf16 MFMAs (v_mfma_f32_16x16x32_bf16)
//                 on 3-way split operands: host-split W3 n-tile resident in 168 VGPRs, P2 split while staged in LDS,
//                 two taps per MFMA, zero halo planes skipped, A fragments by conflict-free ds_read_b128.
//   k_enc_dense1  split-K GEMM [patches,2048]x[2048,208], same arithmetic: 64-row tiles, 8 waves, one workgroup per
//                 CU, host-built weight stages moved by LDS-DMA into two buffers, activations split into LDS.
//   k_enc_head    split-K reduce + bias + tanh + Dense(20) + tanh, one wave per patch, written into each frame's
//                 rows (the patches of several frames can share one launch set: encode_batch_impl).
// All tanh are enc_tanh (v_exp_f32 + v_rcp_f32, |err| < 6e-7).
#include <math.h>

#include "caelo_internal.h"
#include <vector>
#include <stdlib.h>
#include <string.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// tanh x = 1 - 2 / (exp(2x) + 1) on v_exp_f32 + v_rcp_f32: 5 instructions instead of libm's ~50 (the encoder
// evaluates 6.9 M tanh per frame; in k_enc_conv3 alone they were ~8 us).  Absolute error < 6e-7 over the whole
// range (saturates to +-1 through exp -> inf / 0), against a parity budget of 1e-4 on the descriptors.
__device__ inline float enc_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // exp(2x)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// ------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------
#define DENSE_N 200
#define DENSE_NP 208  // padded to 13 MFMA n-tiles
#define DENSE_K 2048
#define C3X_NPAIR 14  // tap pairs of the conv3 kernel (see k_enc_conv3)
#define C3X_TICKET_INT 512   // conv3's eight per-XCD ticket counters: ints 512 + 32 x of the workspace header (bytes 2048 + 128 x)
static void conv3_split_weights(const float *w3, uint4 *out);
static void stage1x_split_weights(const float *w1, const float *w2, uint4 *w1f, uint4 *w2x);
int enc_upload_dense1(const float *wd1, const float *bd1, int K, void **wx_dev, float **bd_dev);
#define HEAD_WQ_FLOATS (13 * 2 * 64 * 4)   // Dense(20) as k_enc_head_mfma's B operand
static void head_weight_fragments(const float *wd2, float *out);

#define S1X_BGO_OFF (512 * 16 + 16)
__global__ void k_enc_bgo_table(float *c0g);

CAELO_API int caelo_set_encoder_weights(caelo_ctx *c, const float *w1, const float *b1, const float *w2,
                                        const float *b2, const float *w3, const float *b3, const float *wd1,
                                        const float *bd1, const float *wd2, const float *bd2) {
    CAELO_REQUIRE(c && w1 && b1 && w2 && b2 && w3 && b3 && wd1 && bd1 && wd2 && bd2, "null argument");
    struct { float **dst; const float *src; size_t n; } plain[] = {
        {&c->enc_w1, w1, 27 * 8}, {&c->enc_b1, b1, 8},   {&c->enc_w2, w2, 27 * 8 * 16}, {&c->enc_b2, b2, 16},
        {&c->enc_w3, w3, 27 * 16 * 32}, {&c->enc_b3, b3, 32}, {&c->enc_wd2, wd2, 200 * 20}, {&c->enc_bd2, bd2, 20}};
    for (auto &p : plain) {
        if (!*p.dst) CAELO_HIP(hipMalloc(p.dst, p.n * sizeof(float)));
        CAELO_HIP(hipMemcpy(*p.dst, p.src, p.n * sizeof(float), hipMemcpyHostToDevice));
    }
    {
        // C0[pos][ch] = b2[ch] + conv2(BG)[pos][ch], BG = tanh(b1) in every valid 8^3 cell, zero padding outside;
        // entries 8192..8199 carry bg itself (the kernel subtracts the same float it was built from)
        float c0[512 * 16 + 8], bg[8];
        for (int ch = 0; ch < 8; ++ch) bg[ch] = tanhf(b1[ch]);
        for (int x = 0; x < 8; ++x) for (int y = 0; y < 8; ++y) for (int z = 0; z < 8; ++z)
            for (int o = 0; o < 16; ++o) {
                float acc = b2[o];
                for (int ka = 0; ka < 3; ++ka) for (int kb = 0; kb < 3; ++kb) for (int kc = 0; kc < 3; ++kc) {
                    const int xx = x + ka - 1, yy = y + kb - 1, zz = z + kc - 1;
                    if (xx < 0 || xx >= 8 || yy < 0 || yy >= 8 || zz < 0 || zz >= 8) continue;
                    for (int ci = 0; ci < 8; ++ci) acc += bg[ci] * w2[(((ka * 3 + kb) * 3 + kc) * 8 + ci) * 16 + o];
                }
                c0[((x * 8 + y) * 8 + z) * 16 + o] = acc;
            }
        for (int ch = 0; ch < 8; ++ch) c0[512 * 16 + ch] = bg[ch];
        // + the pooled outputs of a tile pair that sees nothing but background, per pair and lane of k_enc_stage1x (S1X_BGO_OFF floats
        // in: [16 pairs][64 lanes] float2), computed on the DEVICE with the kernel's own tanh (k_enc_bgo_table)
        if (!c->enc_c0) CAELO_HIP(hipMalloc(&c->enc_c0, (S1X_BGO_OFF + 16 * 64 * 2) * sizeof(float)));
        CAELO_HIP(hipMemcpy(c->enc_c0, c0, sizeof(c0), hipMemcpyHostToDevice));
        k_enc_bgo_table<<<16, 64>>>(c->enc_c0);
        CAELO_LAUNCH_CHECK();
        CAELO_HIP(hipDeviceSynchronize());
    }
    {
        const size_t n1 = 16 * 64, n2 = 18 * 64;  // S1X_W1F_U4, S1X_W2X_U4 (enc_stage1x.inc)
        std::vector<uint4> wxv(n1 + n2);   // (a vector: an early return of CAELO_HIP must not leak the staging buffer)
        uint4 *wx = wxv.data();
        stage1x_split_weights(w1, w2, wx, wx + n1);
        if (!c->enc_w1f) CAELO_HIP(hipMalloc(&c->enc_w1f, n1 * sizeof(uint4)));
        if (!c->enc_w2x) CAELO_HIP(hipMalloc(&c->enc_w2x, n2 * sizeof(uint4)));
        CAELO_HIP(hipMemcpy(c->enc_w1f, wx, n1 * sizeof(uint4), hipMemcpyHostToDevice));
        CAELO_HIP(hipMemcpy(c->enc_w2x, wx + n1, n2 * sizeof(uint4), hipMemcpyHostToDevice));
    }
    {
        const size_t n = (size_t)2 * C3X_NPAIR * 3 * 64;   // (room for three terms; the f16 form fills two)
        std::vector<uint4> wxv(n);
        uint4 *wx = wxv.data();
        conv3_split_weights(w3, wx);
        if (!c->enc_w3x) CAELO_HIP(hipMalloc(&c->enc_w3x, n * sizeof(uint4)));
        CAELO_HIP(hipMemcpy(c->enc_w3x, wx, n * sizeof(uint4), hipMemcpyHostToDevice));
    }
    {
        const int rc = enc_upload_dense1(wd1, bd1, DENSE_K, &c->enc_wd1x, &c->enc_bd1);
        if (rc) return rc;
    }
    {
        const size_t n = (size_t)HEAD_WQ_FLOATS;
        std::vector<float> wqv(n);
        float *wq = wqv.data();
        head_weight_fragments(wd2, wq);
        if (!c->enc_wd2q) CAELO_HIP(hipMalloc(&c->enc_wd2q, n * sizeof(float)));
        CAELO_HIP(hipMemcpy(c->enc_wd2q, wq, n * sizeof(float), hipMemcpyHostToDevice));
    }
    c->has_enc = true;
    return CAELO_OK;
}

// ------------------------------------------------------------------------------------------------
// stage 1: conv1 + pool1 (VALU, sparse) -> conv2 (MFMA, sparse over background) -> pool2 -> P2
// ------------------------------------------------------------------------------------------------
// A voxel patch is mostly empty (2 / 54 / 67 set voxels of 4096 at the three scales), so after
// conv1+pool1 most of the 8^3 cells hold the constant background vector bg = tanh(b1).  conv2 is
// linear before its tanh:  conv2(P1) = conv2(BG) + conv2(P1 - BG),  BG = bg in every valid cell.
//   * C0 = b2 + conv2(BG) is a [512][16] table computed once per model (host, at weight load);
//   * D = P1 - BG is zero except in the few cells whose 4^3 receptive field holds a set voxel.
// The kernel keeps D in LDS (two 4-channel planes with a 1-cell halo), starts every MFMA accumulator
// from C0 and skips -- exactly, a skipped product is an added zero -- every (m-tile, x-tap plane)
// whose 4 x 8 input rows are all-zero in D: 71 % of the conv2 MFMAs on KITTI-shaped scans.
//
// D layout: plane p, padded position q (0..799), channel c:
//   D[p*P1_PLANE + (P1_FRONT + q)*4 + c],   q = (xp*10 + yp)*8 + z,  xp,yp in 0..9 (1-cell halo), z in 0..7
// (no z halo: the two out-of-range z taps are predicated).  An MFMA m-tile is 16 consecutive q, so a
// ds_read_b64 by 32 lanes covers 64 consecutive dwords: bank-conflict free.
#define P1_FRONT 2
#define P1_PLANE ((800 + 2 * P1_FRONT) * 4)

// phase timestamps (100 MHz) of workgroup 0's first patch in the last k_enc_stage1 launch (debug aid)
__device__ unsigned long long g_enc_stamp[16];
#ifdef CAELO_ENC_PROF  // make PROF=1: slots 0..5 accumulate shader-clock cycles per phase over all patches, 6 = patches, 7 = queued cells.
// Caveat: s_memtime drains the wave's outstanding stores first, so the phase that ends with the P2 stores ("conv2")
// is inflated by their latency; use tools/enc_scale_prof.py (unprofiled build) for absolute numbers.
#define ENC_STAMP(i) do { if (threadIdx.x == 0) { const unsigned t_ = (unsigned)clock64(); L.prof[i] += t_ - enc_t_prev; enc_t_prev = t_; } } while (0)
#else
#define ENC_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && patch == 0) g_enc_stamp[i] = wall_clock64(); } while (0)
#endif
int enc_debug_copy(unsigned long long *out_host) {
    CAELO_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_enc_stamp), sizeof(unsigned long long) * 16));
    return CAELO_OK;
}

// k_enc_stage1's D planes carry a z halo as well (z pitch 10): the two out-of-range z taps read stored zeros instead of being
// predicated -- 4 v_cndmask and the s_nops behind them per 6 MFMAs less.  q = (xp*10 + yp)*10 + zp, zp = z + 1.
#define S1P_PLANE ((1000 + 4) * 4)   // floats per 4-channel half, +4 cells: the two halves start 16 banks apart
#define S1_RPITCH 20
#define S1_ROWS (18 * S1_RPITCH)
struct Stage1Lds {
    float p1[2 * S1P_PLANE];
    float w1[27 * 8];
    float b1[8];
    float bg[8];
    unsigned long long cell_mask[512];  // per pooled cell: set voxels of its 4^3 receptive field, bit a*16 + b*4 + j
    unsigned short list_cell[512];      // the cells with a non-zero mask, in arrival order
    unsigned int nzrow[12];  // per padded xp: bit yp set when some cell (xp, yp, *) is non-background
    float2 bgout[4][4][32];  // [wave][yi][lane < 32]: outputs of a tile pair that sees nothing but background (see the kernel)
    unsigned short rows[2][S1_ROWS];  // the patch's 256 voxel rows (16 z bits each) with a zero border: [x + 1][y + 1], pitch S1_RPITCH
    int list_n;
    int next_j;
#ifdef CAELO_ENC_PROF
    unsigned int prof[8];
#endif
};

// the same with a z halo in the planes (k_enc_stage1): no predication
#define CONV2Z_ROW(ACC_A, ACC_B, APTR, KA, KB)                                                     \
    _Pragma("unroll") for (int kc = 0; kc < 3; ++kc) {                                             \
        const int t = (KA) * 9 + (KB) * 3 + kc;                                                     \
        const float2 av = *(const float2 *)((APTR) + (((KA) * 10 + (KB)) * 10 + (kc - 1)) * 4);     \
        ACC_A = MFMA16(av.x, breg[t][0], ACC_A);                                                    \
        ACC_B = MFMA16(av.y, breg[t][1], ACC_B);                                                    \
    }
#define CONV2Z_TILE(ACC_A, ACC_B, APTR, NZ0, NZ1, NZ2)                                              \
    if ((NZ0) & 0x3u) { CONV2Z_ROW(ACC_A, ACC_B, APTR, 0, 0) }                                      \
    if ((NZ0) & 0x6u) { CONV2Z_ROW(ACC_A, ACC_B, APTR, 0, 1) }                                      \
    if ((NZ0) & 0xCu) { CONV2Z_ROW(ACC_A, ACC_B, APTR, 0, 2) }                                      \
    if ((NZ1) & 0x3u) { CONV2Z_ROW(ACC_A, ACC_B, APTR, 1, 0) }                                      \
    if ((NZ1) & 0x6u) { CONV2Z_ROW(ACC_A, ACC_B, APTR, 1, 1) }                                      \
    if ((NZ1) & 0xCu) { CONV2Z_ROW(ACC_A, ACC_B, APTR, 1, 2) }                                      \
    if ((NZ2) & 0x3u) { CONV2Z_ROW(ACC_A, ACC_B, APTR, 2, 0) }                                      \
    if ((NZ2) & 0x6u) { CONV2Z_ROW(ACC_A, ACC_B, APTR, 2, 1) }                                      \
    if ((NZ2) & 0xCu) { CONV2Z_ROW(ACC_A, ACC_B, APTR, 2, 2) }
#define S1_XDEAL(YI) ((YI) == 0 ? 0 : ((YI) == 1 ? 3 : ((YI) == 2 ? 1 : 2)))  // k_enc_stage1's deal of tile pairs to wavefronts
// 3 taps (one (ka, kb) pair of input rows) of one m-tile: 6 MFMAs on two interleaved accumulators
#define CONV2_ROW(ACC_A, ACC_B, APTR, KA, KB)                                                      \
    _Pragma("unroll") for (int kc = 0; kc < 3; ++kc) {                                             \
        const int t = (KA) * 9 + (KB) * 3 + kc;                                                     \
        float2 av = *(const float2 *)((APTR) + (((KA) * 10 + (KB)) * 8 + (kc - 1)) * 4);            \
        if (kc == 0) { if (!zlo) av = make_float2(0.f, 0.f); }                                      \
        if (kc == 2) { if (!zhi) av = make_float2(0.f, 0.f); }                                      \
        ACC_A = MFMA16(av.x, breg[t][0], ACC_A);                                                    \
        ACC_B = MFMA16(av.y, breg[t][1], ACC_B);                                                    \
    }
// all 9 (ka, kb) row pairs of one tile; NZk = occupancy bits of padded plane x + k, shifted down by y0:
// tap row kb reads input rows y0 + kb and y0 + kb + 1
#define CONV2_TILE(ACC_A, ACC_B, APTR, NZ0, NZ1, NZ2)                                               \
    if ((NZ0) & 0x3u) { CONV2_ROW(ACC_A, ACC_B, APTR, 0, 0) }                                       \
    if ((NZ0) & 0x6u) { CONV2_ROW(ACC_A, ACC_B, APTR, 0, 1) }                                       \
    if ((NZ0) & 0xCu) { CONV2_ROW(ACC_A, ACC_B, APTR, 0, 2) }                                       \
    if ((NZ1) & 0x3u) { CONV2_ROW(ACC_A, ACC_B, APTR, 1, 0) }                                       \
    if ((NZ1) & 0x6u) { CONV2_ROW(ACC_A, ACC_B, APTR, 1, 1) }                                       \
    if ((NZ1) & 0xCu) { CONV2_ROW(ACC_A, ACC_B, APTR, 1, 2) }                                       \
    if ((NZ2) & 0x3u) { CONV2_ROW(ACC_A, ACC_B, APTR, 2, 0) }                                       \
    if ((NZ2) & 0x6u) { CONV2_ROW(ACC_A, ACC_B, APTR, 2, 1) }                                       \
    if ((NZ2) & 0xCu) { CONV2_ROW(ACC_A, ACC_B, APTR, 2, 2) }

// Work item J of a launch -> where its bits are and which output row it fills.  J is wave-uniform: everything here
// stays in scalar registers (the per-frame counts come through the scalar cache).
__device__ inline int enc_items_total(const caelo_enc_in &in, int64_t n_patches) {
    if (!in.dedup) return (int)n_patches;
    int acc = 0;
    for (int f = 0; f < in.n_frames; ++f) acc += enc_tables(in, f)->count;
    return acc;
}
__device__ inline int enc_item_row(const caelo_enc_in &in, int J, int &f, int &i) {  // de-duplicated launch
    f = 0;
    i = J;
    while (f + 1 < in.n_frames) {
        const int c = enc_tables(in, f)->count;
        if (i < c) break;
        i -= c;
        ++f;
    }
    return f * in.per_frame + i;
}
__device__ inline void enc_item(const caelo_enc_in &in, int J, int nk, int group, const unsigned long long *&src, int &row) {
    if (in.dedup) {
        int f, i;
        row = enc_item_row(in, J, f, i);
        src = in.bits + (size_t)f * in.frame_stride + (size_t)enc_tables(in, f)->list[i] * 64;
    } else {
        row = (J % nk) * group + (group - 1 - J / nk);  // coarsest scale first
        src = in.bits + (size_t)row * 64;
    }
}

__global__ void __launch_bounds__(256, 3) k_enc_stage1(const caelo_enc_in in, int64_t n_patches,
                                                    int group, int *__restrict__ work_counter,
                                                    const float *__restrict__ w1g, const float *__restrict__ b1g,
                                                    const float *__restrict__ w2g, const float *__restrict__ c0g,
                                                    float *__restrict__ p2out) {
    __shared__ __attribute__((aligned(16))) Stage1Lds L;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = lane >> 4;   // MFMA k-group
    const int n = lane & 15;   // MFMA column (output channel) for B/C, row for A
    // ---- one-time: weights.  B fragment for k-step (tap t, h): W2[t][cin = 2g + h][n]
    float breg[27][2];
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        breg[t][0] = w2g[(t * 8 + 2 * g) * 16 + n];
        breg[t][1] = w2g[(t * 8 + 2 * g + 1) * 16 + n];
    }
    // Tile pairs are dealt to the 4 waves as a XOR Latin square: wave w owns, for every row pair yi, the x pair xp = w ^ d(yi),
    // d = 0, 3, 1, 2 -- every 2x2 cluster of occupied pairs lands on 4 different waves, and of all 24^3 ways to give each wave one
    // pair per yi this one balances the MFMA rows best on the golden frames (tools/conv2_balance.py: average / busiest wavefront
    // 0.83; round 1's xp = (w - 2 yi) mod 4: 0.78; the worst deal: 0.58).
    // C0 = b2 + conv2(BG) accumulator fragments of this lane's 8 m-tiles are patch independent: loaded once,
    // together with the outputs tanh(pool2(C0)) of a pair that sees nothing but background.
    // (the all-background outputs live in LDS, not in registers: with them the kernel spilled 5 registers, and every reload
    // on the background path -- a scratch load followed by s_waitcnt vmcnt(0) -- also waited for the P2 stores before it)
    f32x4 c0r[4][2];
#pragma unroll
    for (int yi = 0; yi < 4; ++yi) {
        const int xp = wave ^ S1_XDEAL(yi);
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
            const float *c0a = c0g + (size_t)((((2 * xp + xt) * 8 + 2 * yi + (g >> 1)) * 8 + 4 * (g & 1)) * 16 + n);
            c0r[yi][xt] = (f32x4){c0a[0], c0a[16], c0a[32], c0a[48]};
        }
        float v0 = fmaxf(fmaxf(c0r[yi][0][0], c0r[yi][0][1]), fmaxf(c0r[yi][1][0], c0r[yi][1][1]));
        float v1 = fmaxf(fmaxf(c0r[yi][0][2], c0r[yi][0][3]), fmaxf(c0r[yi][1][2], c0r[yi][1][3]));
        v0 = fmaxf(v0, __shfl_xor(v0, 32));
        v1 = fmaxf(v1, __shfl_xor(v1, 32));
        if (lane < 32) L.bgout[wave][yi][lane] = make_float2(enc_tanh(v0), enc_tanh(v1));
    }
    for (int i = tid; i < 27 * 8; i += 256) L.w1[i] = w1g[i];
    if (tid < 8) { L.b1[tid] = b1g[tid]; L.bg[tid] = c0g[512 * 16 + tid]; }  // bg = tanh(b1), from the host table
    for (int i = tid; i < 2 * S1P_PLANE; i += 256) L.p1[i] = 0.0f;  // D == 0: halo, pads, background cells
    for (int i = tid; i < 512; i += 256) L.cell_mask[i] = 0ull;
    for (int i = tid; i < 2 * S1_ROWS; i += 256) (&L.rows[0][0])[i] = 0;  // the border stays zero
    if (tid == 0) L.list_n = 0;
    if (tid < 12) L.nzrow[tid] = 0u;
    __syncthreads();
    const int n_items = enc_items_total(in, n_patches);  // distinct patches of the launch (all of them without de-duplication)

#ifdef CAELO_ENC_PROF
    unsigned enc_t_prev = (unsigned)clock64();  // phase totals accumulate in LDS (registers would spill at 3 workgroups/CU)
    if (tid < 8) L.prof[tid] = 0u;
    __syncthreads();
#endif
    // Patches cost ~1x / 1.5x / 2.5x at the three scales (set voxels 2 / 54 / 67).  Work item j -> scale
    // group-1 - j / nk, keypoint j % nk, i.e. coarsest scale first; a persistent workgroup takes the items
    // j = blockIdx.x + k * gridDim.x and so meets a mix of scales instead of one scale only.
    const int nk = (int)(n_patches / group);
    if (blockIdx.x == 0 && tid < 8) work_counter[C3X_TICKET_INT + 32 * tid] = 0;   // conv3 (the next kernel on this stream) draws its pairs from here
    int j = blockIdx.x;
    // thread = one 16-voxel row of the patch (ix = tid >> 4, iy = tid & 15, bit = iz); fetched one patch ahead
    unsigned int row = 0u;
    int patch = 0, patch_next = 0;  // output rows of the current / the prefetched item (wave-uniform)
    if (j < n_items) {
        const unsigned long long *src;
        enc_item(in, j, nk, group, src, patch);
        patch = __builtin_amdgcn_readfirstlane(patch);
        row = ((const unsigned short *)src)[tid];
    }
    const int row_slot = ((tid >> 4) + 1) * S1_RPITCH + (tid & 15) + 1;  // where this thread's row lives in L.rows[*]
    L.rows[0][row_slot] = (unsigned short)row;
    int rbuf = 0;
    __syncthreads();
    while (j < n_items) {
        ENC_STAMP(0);
        // work items beyond the first come from a global counter (patch costs vary 10x: a static split leaves the
        // average workgroup idle for the last ~55 us of the launch); fetched after the mask scatter, published
        // through LDS at the barrier that ends conv1, i.e. microseconds later

        // ---- B1: the receptive-field masks of this thread's two pooled cells (c and c + 256), GATHERED from the 4 x 4 voxel
        // rows that feed them: bit a*16 + b*4 + j <-> voxel (2px-1+a, 2py-1+b, 2pz-1+j).  Two aligned 32-bit LDS reads per
        // (cell, a) = rows y' = 2py .. 2py+3 of the bordered copy; no atomics, no barrier between mask building and queueing
        // (round 1 scattered every set voxel into up to 8 masks with LDS ORs: 1 500 wave instructions per patch, 5 barriers).
        unsigned long long cmask[2];
        {
            const int py = (tid >> 3) & 7, sh = 2 * (tid & 7);
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                const int px = (tid >> 6) + 4 * rep;
                const unsigned int *rp = (const unsigned int *)&L.rows[rbuf][(2 * px) * S1_RPITCH + 2 * py];  // x' = 2px + a, y' = 2py
                unsigned int m32[2] = {0u, 0u};
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const unsigned int w0 = rp[a * (S1_RPITCH / 2)], w1 = rp[a * (S1_RPITCH / 2) + 1];
                    const unsigned int n0 = (((w0 & 0xFFFFu) << 1) >> sh) & 0xFu, n1 = (((w0 >> 16) << 1) >> sh) & 0xFu;
                    const unsigned int n2 = (((w1 & 0xFFFFu) << 1) >> sh) & 0xFu, n3 = (((w1 >> 16) << 1) >> sh) & 0xFu;
                    m32[a >> 1] |= (n0 | (n1 << 4) | (n2 << 8) | (n3 << 12)) << ((a & 1) * 16);
                }
                cmask[rep] = (unsigned long long)m32[0] | ((unsigned long long)m32[1] << 32);
            }
        }
        // issued here, after this patch's prefetched rows were consumed (vmcnt counts in order: an earlier wait for them
        // would also wait for the atomic); consumed at the end of conv1, microseconds later.  Tried and dropped: handing the items
        // out four per atomic (364 -> 348 us for an 8-frame launch, 64 -> 84 us for one frame), and fetching one patch AHEAD (the
        // round trip then never shows in the phase profile, but pinning a second item per workgroup costs balance: 359 -> 365 us
        // per 8-frame launch, 62 -> 66 us for one frame, 11.09 -> 11.03 k frames/s).
        int j_fetch;  // defined in thread 0 only, and only read there (no merge copy that would wait for the atomic)
        if (tid == 0) j_fetch = atomicAdd(work_counter, 1);
        ENC_STAMP(1);
        // ---- queue the cells with a non-empty mask (one LDS counter bump per wave)
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int cell = tid + rep * 256;
            const bool hit = cmask[rep] != 0ull;
            const unsigned long long bal = __ballot(hit);
            if (bal != 0ull) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&L.list_n, __popcll(bal));
                base = __shfl(base, 0);
                if (hit) {
                    L.list_cell[base + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)cell;
                    L.cell_mask[cell] = cmask[rep];
                    atomicOr(&L.nzrow[(cell >> 6) + 1], 1u << (((cell >> 3) & 7) + 1));
                }
            }
        }
        caelo_lds_barrier();
        ENC_STAMP(2);
        const int nlist = L.list_n;
        // ---- B2: conv1 + pool1 + tanh on the queued cells; 8 lanes = the 8 positions of a pooling block
        for (int base = wave * 8; base < nlist; base += 32) {
            const int item = base + (lane >> 3);
            const int sub = lane & 7;
            float acc[8];
            int cell = 0;
            if (item < nlist) {
                cell = L.list_cell[item];
                const unsigned long long mask = L.cell_mask[cell];
                const int sa = sub >> 2, sb = (sub >> 1) & 1, sc = sub & 1;
                unsigned int taps = 0;  // bit (ka*3+kb)*3+kc
#pragma unroll
                for (int ka = 0; ka < 3; ++ka)
#pragma unroll
                    for (int kb = 0; kb < 3; ++kb)
                        taps |= ((unsigned int)(mask >> ((sa + ka) * 16 + (sb + kb) * 4 + sc)) & 7u) << ((ka * 3 + kb) * 3);
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = L.b1[c];
                while (taps) {  // ascending tap order == the oracle's (kx,ky,kz) order
                    const int t = __ffs((int)taps) - 1;
                    taps &= taps - 1;
                    const float4 wa = *(const float4 *)&L.w1[t * 8], wb = *(const float4 *)&L.w1[t * 8 + 4];
                    acc[0] += wa.x; acc[1] += wa.y; acc[2] += wa.z; acc[3] += wa.w;
                    acc[4] += wb.x; acc[5] += wb.y; acc[6] += wb.z; acc[7] += wb.w;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = -3.0e38f;
            }
            // max over the pooling block = 8 aligned lanes (tanh is monotone: pool the pre-activations); DPP
            // lane swaps (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror) instead of 24 ds_bpermute round trips
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                acc[c] = fmaxf(acc[c], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[c]), 0xB1, 0xF, 0xF, true)));
                acc[c] = fmaxf(acc[c], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[c]), 0x4E, 0xF, 0xF, true)));
                acc[c] = fmaxf(acc[c], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[c]), 0x141, 0xF, 0xF, true)));
            }
            if (item < nlist) {
                float mine = acc[0];  // lane `sub` finishes channel `sub`
#pragma unroll
                for (int c = 1; c < 8; ++c) mine = (sub == c) ? acc[c] : mine;
                const int px = cell >> 6, py = (cell >> 3) & 7, pz = cell & 7;
                const int q = ((px + 1) * 10 + (py + 1)) * 10 + pz + 1;
                L.p1[(sub >> 2) * S1P_PLANE + q * 4 + (sub & 3)] = enc_tanh(mine) - L.bg[sub];
            }
        }
        if (tid == 0) L.next_j = (int)gridDim.x + j_fetch;
        caelo_lds_barrier();
        ENC_STAMP(3);
        const int jn = __builtin_amdgcn_readfirstlane(L.next_j);
        unsigned int row_next = 0u;
        if (jn < n_items) {
            const unsigned long long *src;
            enc_item(in, jn, nk, group, src, patch_next);
            patch_next = __builtin_amdgcn_readfirstlane(patch_next);
            row_next = ((const unsigned short *)src)[tid];
        }
        // ---- conv2 (8->16) on MFMA: per row pair yi this wave owns the x pair xp = wave ^ d(yi)
        {
            const int yl = n >> 3, z = n & 7;  // A row m = yl*8 + z
            const float *plane = L.p1 + (g >> 1) * S1P_PLANE + 2 * (g & 1);
#pragma unroll
            for (int yi = 0; yi < 4; ++yi) {
                const int y0 = 2 * yi;
                const int xp = wave ^ S1_XDEAL(yi);
                // wave-uniform occupancy of the input rows yp = y0 .. y0+3 in the padded planes 2xp .. 2xp+3
                const unsigned int r0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((L.nzrow[2 * xp] >> y0) & 0xFu));
                const unsigned int r1 = (unsigned)__builtin_amdgcn_readfirstlane((int)((L.nzrow[2 * xp + 1] >> y0) & 0xFu));
                const unsigned int r2 = (unsigned)__builtin_amdgcn_readfirstlane((int)((L.nzrow[2 * xp + 2] >> y0) & 0xFu));
                const unsigned int r3 = (unsigned)__builtin_amdgcn_readfirstlane((int)((L.nzrow[2 * xp + 3] >> y0) & 0xFu));
                // C column = n (channel), rows 4g..4g+3 -> z = 4*(g&1)+r ; pooled cell (xp, yi, 2*(g&1)+{0,1})
                float *dst = p2out + (size_t)patch * 1024 + (size_t)(((xp * 4 + yi) * 4 + 2 * (g & 1)) * 16 + n);
                if ((r0 | r1 | r2 | r3) == 0u) {  // nothing but background feeds this pair: per-model constants
                    if (g < 2) {
                        const float2 o = L.bgout[wave][yi][lane];
                        dst[0] = o.x; dst[16] = o.y;
                    }
                    continue;
                }
                // accumulators start from C0 = b2 + conv2(BG); two per tile to keep the MFMA chains independent
                f32x4 acc0 = c0r[yi][0];
                f32x4 acc1 = c0r[yi][1];
                f32x4 acc0b = {0.f, 0.f, 0.f, 0.f}, acc1b = {0.f, 0.f, 0.f, 0.f};
                // padded position of (x = 2xp, y = y0 + yl, z) for tap (0,0,1): xp' = x + ka, yp = y + kb
                const int qbase = ((2 * xp) * 10 + (y0 + yl)) * 10 + z + 1;
                const float *a0 = plane + qbase * 4;
                const float *a1 = a0 + 100 * 4;
                CONV2Z_TILE(acc0, acc0b, a0, r0, r1, r2)
                CONV2Z_TILE(acc1, acc1b, a1, r1, r2, r3)
                acc0 += acc0b;
                acc1 += acc1b;
                // ---- pool2: x pair in registers, z pairs in registers, y pair across lanes g <-> g^2
                float v0 = fmaxf(fmaxf(acc0[0], acc0[1]), fmaxf(acc1[0], acc1[1]));  // pz = 2*(g&1)
                float v1 = fmaxf(fmaxf(acc0[2], acc0[3]), fmaxf(acc1[2], acc1[3]));  // pz = 2*(g&1)+1
                v0 = fmaxf(v0, __shfl_xor(v0, 32));
                v1 = fmaxf(v1, __shfl_xor(v1, 32));
                if (g < 2) {
                    dst[0] = enc_tanh(v0);
                    dst[16] = enc_tanh(v1);
                }
            }
        }
        caelo_lds_barrier();
        ENC_STAMP(4);
#ifdef CAELO_ENC_PROF
        if (threadIdx.x == 0) { L.prof[6] += 1; L.prof[7] += nlist; }
#else
        if (threadIdx.x == 0 && blockIdx.x == 0 && patch == 0) g_enc_stamp[8] = (unsigned long long)nlist;
#endif
        // ---- back to D == 0 for the next patch
        for (int i = tid; i < nlist * 2; i += 256) {
            const int cell = L.list_cell[i >> 1];
            const int px = cell >> 6, py = (cell >> 3) & 7, pz = cell & 7;
            const int q = ((px + 1) * 10 + (py + 1)) * 10 + pz + 1;
            *(float4 *)&L.p1[(i & 1) * S1P_PLANE + q * 4] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid == 0) L.list_n = 0;
        if (tid < 12) L.nzrow[tid] = 0u;
        L.rows[rbuf ^ 1][row_slot] = (unsigned short)row_next;  // the next patch's rows (fetched during conv2)
        rbuf ^= 1;
        caelo_lds_barrier();
        j = jn;
        row = row_next;
        patch = patch_next;
        ENC_STAMP(5);
    }
#ifdef CAELO_ENC_PROF
    if (threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_enc_stamp[i], (unsigned long long)L.prof[i]);
#endif
}

#include "enc_stage1x.inc"

#define ENC_XCD_CSTRIDE 32   // ints between two per-XCD work counters (one 128-byte line each)
// ------------------------------------------------------------------------------------------------
// conv3 (16->32) implicit GEMM: M = 64 positions/patch, N = 32, K = 27*16 -- f32 products on the bf16 matrix pipe
// ------------------------------------------------------------------------------------------------
// v_mfma_f32_16x16x32_bf16 retires 16x the FLOPs per cycle of the f32-input MFMA (MI355X_MICROARCH: 2.5 PF vs
// 157 TF).  An f32 value splits EXACTLY into three bf16 terms, x = hi + mid + lo with hi = bf16(x),
// mid = bf16(x - hi), lo = bf16(x - hi - mid) (both differences are exact in f32, |x - hi - mid - lo| <= 2^-27 |x|),
// and a product a*b is a_hi b_hi + a_hi b_mid + a_mid b_hi + a_mid b_mid + a_hi b_lo + a_lo b_hi up to 2^-27 |a b|
// -- below the 2^-24 rounding of an f32 multiply.  Every partial product of two bf16 is exact in f32 and the MFMA
// accumulates in f32: the result is f32-grade (measured against the oracle in tests/) at 6/16 of the f32 MFMA's time.
//
// The weights are split once on the host (caelo_set_encoder_weights -> enc_w3x, already in register order), the
// activations once per patch while P2 is staged in LDS.  LDS per patch slot: [split 3][channel half 2][pos][8] bf16
// (16 B per position), pos = (xp*6 + yp)*8 + zp over the zero-haloed 6x6x6 volume (z pitch 8).  K = 32 per MFMA =
// two taps x 16 channels: lanes g < 2 carry tap A of a pair, lanes g >= 2 tap B; taps are paired so that B - A is
// one of three constant position offsets (+1 z, +1 y, +1 x), which the upper lanes fold into their base address --
// every other offset is an instruction immediate.  An m-tile is one x plane (lane m -> y = m >> 2, z = m & 3); with
// the z pitch of 8 and the second channel half displaced by 4 positions, each 16-lane group of a ds_read_b128
// (MI355X_MICROARCH LDS table) touches all 64 banks once.
#define C3X_PLANE 48                 // positions per padded x plane (6 rows of 8)
#define C3X_ARR 292                  // positions per (split, half) array: 288 used, = 4 mod 16 (bank rotation of half 1)
#define C3X_SPLIT (2 * C3X_ARR)
// Round 3: the f32 products of conv3 run as TWO f16 terms per operand (x = hi + lo, |x - hi - lo| <= max(2^-22 |x|, 2^-25); the MFMA honours
// f16 subnormals, tools/micro/f16_mfma_subnormal.hip) and three partial products hi hi + hi lo + lo hi -- 3 MFMAs per K = 32 slab
// instead of the 6 of the 3-way bf16 split (C3X_F16=0, round 2), 112 instead of 168 B registers, two LDS planes instead of three.
// The dropped lo lo term is 2^-22 relative; measured against the f32 oracle the descriptors do not move (tools/enc_layer_errors.py).
#ifndef C3X_F16
#define C3X_F16 1
#endif
#define C3X_NS (C3X_F16 ? 2 : 3)     // terms per operand
#define C3X_SLOT (C3X_NS * C3X_SPLIT)     // positions (x 16 B) per patch slot
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// tap pairs (A, B): 9 x (kc 0,1) | 3 x (kb 0,1 at kc 2) | (ka 0,1 at kb 2, kc 2) | tap 26 alone
__host__ __device__ constexpr int c3x_pairA(int p) { return p < 9 ? p * 3 : (p < 12 ? (p - 9) * 9 + 2 : (p == 12 ? 8 : 26)); }
__host__ __device__ constexpr int c3x_pairB(int p) { return p < 9 ? p * 3 + 1 : (p < 12 ? (p - 9) * 9 + 5 : (p == 12 ? 17 : -1)); }
__host__ __device__ constexpr int c3x_pairClass(int p) { return p < 9 ? 0 : (p < 12 ? 1 : (p == 12 ? 2 : 3)); }
__host__ __device__ constexpr int c3x_tapPos(int t) { return (t / 9) * C3X_PLANE + ((t / 3) % 3) * 8 + (t % 3); }
// x plane X never sees data through the taps of pair P (all of them read the zero halo plane): skipped exactly
__host__ __device__ constexpr bool c3x_dead(int x, int p) {
    return (x == 0 && c3x_pairA(p) / 9 == 0 && (c3x_pairB(p) < 0 || c3x_pairB(p) / 9 == 0)) ||
           (x == 3 && c3x_pairA(p) / 9 == 2 && (c3x_pairB(p) < 0 || c3x_pairB(p) / 9 == 2));
}

__host__ __device__ inline uint32_t enc_bf16_rne(float x) {  // bf16(x), round to nearest even, as the high half of an f32 pattern
    union { float f; uint32_t u; } v = {x};
    return (v.u + 0x7FFFu + ((v.u >> 16) & 1u)) & 0xFFFF0000u;
}
__host__ __device__ inline float enc_bits_f32(uint32_t u) {
    union { uint32_t u; float f; } v = {u};
    return v.f;
}
__host__ __device__ inline void enc_split3(float x, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
    hi = enc_bf16_rne(x);
    const float r = x - enc_bits_f32(hi);
    mid = enc_bf16_rne(r);
    lo = enc_bf16_rne(r - enc_bits_f32(mid));
}
// device: two values at once on the hardware converter (v_cvt_pk_bf16_f32, round to nearest even like enc_bf16_rne: the same
// bits for every finite input, 5.5 instead of 12 VALU instructions per value).  Result words hold value 0 in the low half.
__device__ inline void enc_split3_pk(float x0, float x1, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
    typedef __bf16 enc_bf2 __attribute__((ext_vector_type(2)));
    typedef float enc_f2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector((enc_f2){x0, x1}, enc_bf2));
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xFFFF0000u);
    mid = __builtin_bit_cast(uint32_t, __builtin_convertvector((enc_f2){r0, r1}, enc_bf2));
    const float q0 = r0 - __uint_as_float(mid << 16), q1 = r1 - __uint_as_float(mid & 0xFFFF0000u);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector((enc_f2){q0, q1}, enc_bf2));
}
// two values -> their f16 (hi, lo) terms, packed (value 0 in the low half); hi = RNE f16(x), lo = RNE f16(x - hi)
__device__ inline void enc_split2h_pk(float x0, float x1, uint32_t &hi, uint32_t &lo) {
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
    hi = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
    lo = (uint32_t)__builtin_bit_cast(uint16_t, l0) | ((uint32_t)__builtin_bit_cast(uint16_t, l1) << 16);
}
#define ENC_PK8(A) make_uint4((A[0] >> 16) | A[1], (A[2] >> 16) | A[3], (A[4] >> 16) | A[5], (A[6] >> 16) | A[7])

// host: W3 [27][16][32] -> [ntile 2][pair 14][split 3][lane 64] uint4, the B operand of lane (n = lane & 15, g = lane >> 4)
static void conv3_split_weights(const float *w3, uint4 *out) {
    for (int nt = 0; nt < 2; ++nt)
        for (int p = 0; p < C3X_NPAIR; ++p)
            for (int lane = 0; lane < 64; ++lane) {
                const int n = lane & 15, g = lane >> 4;
                const int tap = g < 2 ? c3x_pairA(p) : c3x_pairB(p);
                uint4 *o = out + ((size_t)(nt * C3X_NPAIR + p) * C3X_NS) * 64 + lane;
#if C3X_F16
                uint16_t h[8], l[8];
                for (int i = 0; i < 8; ++i)
                    enc_split2h(tap >= 0 ? w3[(tap * 16 + 8 * (g & 1) + i) * 32 + nt * 16 + n] : 0.0f, h[i], l[i]);
                o[0] = make_uint4(h[0] | (uint32_t)h[1] << 16, h[2] | (uint32_t)h[3] << 16, h[4] | (uint32_t)h[5] << 16, h[6] | (uint32_t)h[7] << 16);
                o[64] = make_uint4(l[0] | (uint32_t)l[1] << 16, l[2] | (uint32_t)l[3] << 16, l[4] | (uint32_t)l[5] << 16, l[6] | (uint32_t)l[7] << 16);
#else
                uint32_t h[8], m[8], l[8];
                for (int i = 0; i < 8; ++i)
                    enc_split3(tap >= 0 ? w3[(tap * 16 + 8 * (g & 1) + i) * 32 + nt * 16 + n] : 0.0f, h[i], m[i], l[i]);
                o[0] = ENC_PK8(h);
                o[64] = ENC_PK8(m);
                o[128] = ENC_PK8(l);
#endif
            }
}

#ifndef C3X_WGS
#define C3X_WGS 2   // workgroups per CU the register budget is set for (3 spill at 168 registers: 183 instead of 92 us per 8 frames)
#endif
__global__ void __launch_bounds__(256, C3X_WGS) k_enc_conv3(const float *__restrict__ p2, int64_t n_patches, const caelo_enc_in in,
                                                      const uint4 *__restrict__ w3x, const float *__restrict__ b3g,
                                                      float *__restrict__ f3, int *__restrict__ stage1_counter,
                                                      int *__restrict__ xcd_counters) {
    __shared__ uint4 S[2 * C3X_SLOT];
    // stage 1 (the previous kernel on this stream) is complete: hand its work counter back at zero, so that no
    // memset launch sits on the encoder stream's critical path
    if (blockIdx.x == 0 && threadIdx.x == 0) *stage1_counter = 0;
    if (blockIdx.x == 0 && threadIdx.x < 8) xcd_counters[threadIdx.x * ENC_XCD_CSTRIDE] = 0;  // stage 1's per-XCD queues, for the next launch set
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    const int ntile = wave & 1, slot = __builtin_amdgcn_readfirstlane(wave >> 1);
    uint4 bq[C3X_NPAIR][C3X_NS];
#pragma unroll
    for (int p = 0; p < C3X_NPAIR; ++p)
#pragma unroll
        for (int sp = 0; sp < C3X_NS; ++sp) bq[p][sp] = w3x[((size_t)(ntile * C3X_NPAIR + p) * C3X_NS + sp) * 64 + lane];
    const float bias = b3g[16 * ntile + n];
    for (int i = tid; i < 2 * C3X_SLOT; i += 256) S[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const int n_items = enc_items_total(in, n_patches);
    const int n_pairs = (n_items + 1) / 2;
    // item (wave-uniform) -> row of P2 / F3 (the row stage 1 wrote)
#define C3_ROW(ITEM, ROW)                                     \
    {                                                         \
        ROW = (ITEM);                                         \
        if (in.dedup) {                                       \
            int f_, i_;                                       \
            ROW = enc_item_row(in, (ITEM), f_, i_);           \
        }                                                     \
    }
    // register staging: the next pair's 2 x 4 KB are fetched while this pair's MFMAs run; thread -> (slot, position,
    // channel half): 8 consecutive channels
    const int f_slot = __builtin_amdgcn_readfirstlane(tid >> 7), f_pos = (tid >> 1) & 63, f_h = tid & 1;
    const int f_q = (((f_pos >> 4) + 1) * 6 + ((f_pos >> 2) & 3) + 1) * 8 + (f_pos & 3) + 1;
    float4 pre0, pre1;
#define C3_FETCH(PAIR)                                                                           \
    {                                                                                            \
        const int pa_ = (PAIR) * 2 + f_slot;                                                     \
        pre0 = make_float4(0.f, 0.f, 0.f, 0.f);                                                  \
        pre1 = pre0;                                                                             \
        if ((PAIR) < n_pairs && pa_ < n_items) {                                                 \
            int row_;                                                                            \
            C3_ROW(pa_, row_)                                                                    \
            const float4 *q_ = (const float4 *)(p2 + (size_t)row_ * 1024 + f_pos * 16 + 8 * f_h); \
            pre0 = q_[0];                                                                        \
            pre1 = q_[1];                                                                        \
        }                                                                                        \
    }
#ifdef CAELO_C3_PROF   // make C3PROF=1 (NOT together with PROF=1: stage 1's profile uses the same slots of g_enc_stamp)
    unsigned long long c3_t[4] = {0ull, 0ull, 0ull, 0ull};
    unsigned c3_prev = (unsigned)clock64();
#define C3_STAMP(i) do { if (tid == 0) { const unsigned t_ = (unsigned)clock64(); c3_t[i] += t_ - c3_prev; c3_prev = t_; } } while (0)
#else
#define C3_STAMP(i) do { } while (0)
#endif
    // Work distribution: eight per-XCD queues as in stage 1 (pair p belongs to XCD p % 8; ONE counter for the 12 288 pairs of an
    // 8-frame launch would retire its same-address atomics in ~150 us).  A workgroup's first two queue positions are static (its
    // rank among the XCD's workgroups, then rank + their number), later ones are tickets drawn one iteration before they are needed
    // as `next` (stage 1 zeroes the counters).  Inside the frame pipeline the grid leaves a quarter of the slots free and the other
    // streams' kernels slow the workgroups of SOME CUs: with the static stride every workgroup did the same number of pairs and the
    // launch ended with its slowest CU.
    __shared__ int s_after[2];   // by iteration parity: thread 0 may be an iteration ahead of the slowest reader
    const int xcd = (int)(blockIdx.x & 7u), wgs_of_xcd = ((int)gridDim.x - xcd + 7) >> 3;
    int *const ticket = stage1_counter + C3X_TICKET_INT + 32 * xcd;
    int pair = xcd + 8 * (int)(blockIdx.x >> 3), next = pair + 8 * wgs_of_xcd;
    int drawn = 0;   // thread 0: the ticket drawn in the previous iteration
    if (tid == 0) drawn = atomicAdd(ticket, 1);
    C3_FETCH(pair)
    for (int it = 0; pair < n_pairs; pair = next, next = __builtin_amdgcn_readfirstlane(s_after[it & 1]), ++it) {
        C3_STAMP(3);
        {   // split the staged 8 channels into the three bf16 terms: 3 x 16 B into LDS
            const float v[8] = {pre0.x, pre0.y, pre0.z, pre0.w, pre1.x, pre1.y, pre1.z, pre1.w};
            uint4 *d = &S[f_slot * C3X_SLOT + f_h * C3X_ARR + f_q];
#if C3X_F16
            uint32_t h[4], l[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) enc_split2h_pk(v[2 * k], v[2 * k + 1], h[k], l[k]);
            d[0] = make_uint4(h[0], h[1], h[2], h[3]);
            d[C3X_SPLIT] = make_uint4(l[0], l[1], l[2], l[3]);
#else
            uint32_t h[4], m[4], l[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) enc_split3_pk(v[2 * k], v[2 * k + 1], h[k], m[k], l[k]);
            d[0] = make_uint4(h[0], h[1], h[2], h[3]);
            d[C3X_SPLIT] = make_uint4(m[0], m[1], m[2], m[3]);
            d[2 * C3X_SPLIT] = make_uint4(l[0], l[1], l[2], l[3]);
#endif
        }
        if (tid == 0) s_after[it & 1] = xcd + 8 * (2 * wgs_of_xcd + drawn);   // the pair after `next` (read after the loop's second barrier)
        C3_STAMP(0);
        __syncthreads();
        C3_STAMP(1);
        if (tid == 0) drawn = atomicAdd(ticket, 1);
        C3_FETCH(next)
        const int item = pair * 2 + slot;
        int patch;
        C3_ROW(item, patch)
        // lane (m = n, g): output (y = n >> 2, z = n & 3) of x plane X reads padded (X + ka, y + kb, z + kc)
        const uint4 *base = &S[slot * C3X_SLOT + (g & 1) * C3X_ARR + (n >> 2) * 8 + (n & 3)];
        const uint4 *ab[4] = {base + (g >= 2 ? 1 : 0), base + (g >= 2 ? 8 : 0), base + (g >= 2 ? C3X_PLANE : 0), base};
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            f32x4 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = (f32x4){bias, bias, bias, bias};
#pragma unroll
            for (int p = 0; p < C3X_NPAIR; ++p) {
#if C3X_F16
                typedef _Float16 c3_h8 __attribute__((ext_vector_type(8)));
                const c3_h8 bh = __builtin_bit_cast(c3_h8, bq[p][0]), bl = __builtin_bit_cast(c3_h8, bq[p][1]);
                c3_h8 ah[2], al[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (c3x_dead(2 * round + j, p)) continue;
                    const uint4 *a = ab[c3x_pairClass(p)] + c3x_tapPos(c3x_pairA(p)) + (2 * round + j) * C3X_PLANE;
                    ah[j] = __builtin_bit_cast(c3_h8, a[0]);
                    al[j] = __builtin_bit_cast(c3_h8, a[C3X_SPLIT]);
                }
                // smallest terms first; the two accumulators alternate so that no MFMA waits for its predecessor
#define C3X_MAC(A, B)                                                                                           \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) if (!c3x_dead(2 * round + j, p))                              \
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[j], B, acc[j], 0, 0, 0);
                C3X_MAC(al, bh)
                C3X_MAC(ah, bl)
                C3X_MAC(ah, bh)
#else
                const bf16x8 bh = __builtin_bit_cast(bf16x8, bq[p][0]), bm = __builtin_bit_cast(bf16x8, bq[p][1]),
                             bl = __builtin_bit_cast(bf16x8, bq[p][2]);
                bf16x8 ah[2], am[2], al[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (c3x_dead(2 * round + j, p)) continue;
                    const uint4 *a = ab[c3x_pairClass(p)] + c3x_tapPos(c3x_pairA(p)) + (2 * round + j) * C3X_PLANE;
                    ah[j] = __builtin_bit_cast(bf16x8, a[0]);
                    am[j] = __builtin_bit_cast(bf16x8, a[C3X_SPLIT]);
                    al[j] = __builtin_bit_cast(bf16x8, a[2 * C3X_SPLIT]);
                }
                // smallest terms first; the two accumulators alternate so that no MFMA waits for its predecessor
#define C3X_MAC(A, B)                                                                                           \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) if (!c3x_dead(2 * round + j, p))                              \
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[j], B, acc[j], 0, 0, 0);
                C3X_MAC(al, bh)
                C3X_MAC(ah, bl)
                C3X_MAC(am, bm)
                C3X_MAC(am, bh)
                C3X_MAC(ah, bm)
                C3X_MAC(ah, bh)
#endif
            }
            if (item < n_items) {
                // C rows 4g + r -> (y = g, z = r); flatten index (x,y,z,c) = ((x*4 + y)*4 + z)*32 + c
                float *dst = f3 + (size_t)patch * 2048 + 16 * ntile + n;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[(((2 * round + j) * 4 + g) * 4 + r) * 32] = enc_tanh(acc[j][r]);
            }
        }
        C3_STAMP(2);
        __syncthreads();
    }
#ifdef CAELO_C3_PROF
    if (tid == 0) {
        for (int i = 0; i < 4; ++i) atomicAdd(&g_enc_stamp[8 + i], c3_t[i]);
        atomicAdd(&g_enc_stamp[12], 1ull);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// dense1: split-K GEMM  part[s][row][208] = F3[row][k0:k0+KTOT/8] x Wd1[k0:k0+KTOT/8][208] -- f16 x 2 like conv3
// ------------------------------------------------------------------------------------------------
#define D1_BM 64
#define D1_BK 32      // one v_mfma_f32_16x16x32_f16 k-step per stage
#define D1_SPLIT 8  // the most k slices any instance uses: sizes the partial-sum buffer
// k slices per row tile: 8 for the long-K instance (K = 16384: 16 row tiles x 8 = 128 workgroups per frame); 4 for K = 2048 --
// one frame still fills the chip (48 x 4 workgroups, 29 us either way), a batch of 8 frames writes and re-reads half the partial
// sums (10.3 -> 10.6 k frames/s; 2 slices: 10.7 k but 48 us for a single frame).  One value for every launch size: the head adds
// the slices in order, so the descriptors' low bits depend on it and single calls must equal batched ones bit for bit.
#ifndef D1_SPLIT_2048
#define D1_SPLIT_2048 2   // k slices of Dense(200) (K = 2048).  Round 6: 4 -> 2 (half the partial sums written and read back by the head: head 22 -> 15 us, Dense(200) 83 -> 90 us per 24 576 rows, +1.5-3 % frames/s in 11 of 11 A/B runs, profiles/r06_dense1_split.txt; `make EXTRA=-DD1_SPLIT_2048=4 ...` builds the old one)
#endif
#define D1_SPLIT_OF(KTOT) ((KTOT) > 2048 ? 8 : D1_SPLIT_2048)
#define D1_THREADS 512
#define D1_NT (DENSE_NP / 16)          // 13 n-tiles
// Round 3: like conv3, Dense(200) evaluates its f32 products from TWO f16 terms per operand and three partial products
// (hi hi + hi lo + lo hi; the dropped lo lo term is 2^-22 relative): 3 MFMAs per K = 32 slab instead of the 6 of the 3-way bf16
// split, a 26 KB instead of a 39 KB weight stage (the LDS-DMA weight stream was this kernel's ceiling), two A planes instead of three.
#define D1_NS 2                        // terms per operand
#define D1_B16 (D1_NS * D1_NT * 64)    // uint4 of a B stage: [term][n-tile][lane]
#define D1_BSLOTS ((D1_B16 + D1_THREADS - 1) / D1_THREADS)

// host: Wd1 [K][200] -> per 32-deep k-step the LDS image of the B operand: [kstep][f16 term 2][n-tile 13][lane 64] x 16 B,
// lane (n = lane & 15, g = lane >> 4) holding k = 32 kstep + 8 g .. + 7 of column 16 ntile + n (zero beyond 200)
static void dense1_split_weights(const float *wd1, int K, uint4 *out) {
    for (int ks = 0; ks < K / 32; ++ks)
        for (int nt = 0; nt < D1_NT; ++nt)
            for (int lane = 0; lane < 64; ++lane) {
                const int n = lane & 15, g = lane >> 4, col = nt * 16 + n;
                uint16_t h[8], l[8];
                for (int i = 0; i < 8; ++i)
                    enc_split2h(col < DENSE_N ? wd1[(size_t)(ks * 32 + 8 * g + i) * DENSE_N + col] : 0.0f, h[i], l[i]);
                uint4 *o = out + (size_t)ks * D1_B16 + nt * 64 + lane;
                o[0] = make_uint4(h[0] | (uint32_t)h[1] << 16, h[2] | (uint32_t)h[3] << 16, h[4] | (uint32_t)h[5] << 16, h[6] | (uint32_t)h[7] << 16);
                o[D1_NT * 64] = make_uint4(l[0] | (uint32_t)l[1] << 16, l[2] | (uint32_t)l[3] << 16, l[4] | (uint32_t)l[5] << 16, l[6] | (uint32_t)l[7] << 16);
            }
}

int enc_upload_dense1(const float *wd1, const float *bd1, int K, void **wx_dev, float **bd_dev) {
    const size_t n = (size_t)(K / 32) * D1_B16;
    uint4 *wx = (uint4 *)malloc(n * sizeof(uint4));
    if (!wx) { caelo_set_error("out of host memory"); return CAELO_ERR_ARG; }
    dense1_split_weights(wd1, K, wx);
    float bd[DENSE_NP] = {0.0f};
    memcpy(bd, bd1, DENSE_N * sizeof(float));
    hipError_t e = hipSuccess;
    if (!*wx_dev) e = hipMalloc(wx_dev, (n + 6 * 64) * sizeof(uint4));  // + six 1 KB pieces: k_enc_dense1p's waves copy 32 pieces per 26-piece stage
    if (e == hipSuccess && !*bd_dev) e = hipMalloc((void **)bd_dev, sizeof(bd));
    if (e == hipSuccess) e = hipMemcpy(*wx_dev, wx, n * sizeof(uint4), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(*bd_dev, bd, sizeof(bd), hipMemcpyHostToDevice);
    free(wx);
    if (e != hipSuccess) { caelo_set_error("dense_1 upload failed: %s", hipGetErrorString(e)); return CAELO_ERR_HIP; }
    return CAELO_OK;
}

typedef _Float16 d1_h8 __attribute__((ext_vector_type(8)));
#define D1_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_f16((A), (B), (C), 0, 0, 0)

// ------------------------------------------------------------------------------------------------
// dense1, software-pipelined.  k_enc_dense1 above needs 90 KB of LDS, so ONE workgroup runs per CU, and its stage is a chain --
// write A, barrier, fetch fragments from LDS, 42 MFMAs per wave, barrier -- in which the matrix pipe idles through every LDS
// round trip (SQ counters, 8-frame launch: MFMA busy 35 %, waves parked at waitcnt / barrier 51 % of their cycles).  This kernel
// keeps the one workgroup per CU and gives it the whole LDS: A double-buffered, THREE weight stages, so that
//   - the fragments of the next group of column tiles (and of the next stage, across the barrier) are in flight while the
//     current group's MFMAs run: the pipe has work on both sides of the single barrier of a stage;
//   - the weight DMA runs two stages ahead and the rows of F3 (HBM) two stages ahead in alternating registers; the barrier
//     waits with vmcnt(one stage's loads): only the older stage must have landed.
// The weight DMA, the F3 loads and the A stores are inline asm with hand-counted waits: with a compiler-visible LDS-DMA in the
// loop hipcc waits vmcnt(0) before every use of a loaded register and before every LDS store, and lgkmcnt(0) before every use
// of an LDS fragment (measured on a three-line kernel); without one its lgkmcnt arithmetic is exact.  Same products in the same order per accumulator as k_enc_dense1: bit-identical partial sums.
#define D1P_BSTRIDE (26 * 64)  // uint4 per weight buffer: the 26 pieces of 1 KB of a stage (waves 6 and 7 copy pieces 22..25, twice over: every wave issues
                               // four loads per stage, which the vmcnt arithmetic relies on, and nothing lands outside the buffer -- round 3's buffers
                               // were 32 KB for the six pieces the last waves brought along: 128 KB of LDS for the 128-row instance, now 110)
#define D1P_NBUF(MTW) 3
#define D1P_LDS_BYTES(MTW) ((2 * D1_NS * 4 * D1_BM * (MTW) + D1P_NBUF(MTW) * D1P_BSTRIDE) * 16)
// MTW = 1: 64 rows per workgroup, 94 KB.  MTW = 2: 128 rows, 110 KB -- every wave owns two row tiles, each weight fragment feeds
// two MFMAs, and the weight stream from L2 halves.  History of the 8-frame launch (24 576 rows): bf16 x 3 with 39 KB stages
// streamed 983 MB at 64 rows (125 us = 7.9 TB/s, the LDS-DMA ceiling of the chip) and 118 us at 128 rows with TWO weight buffers
// (all the LDS there was); f16 x 2 (26 KB stages, half the MFMAs) 101 us with two buffers -- a stage then cannot be shorter than
// one L2 -> LDS round trip, ~2 us under this load, whatever the matrix pipe does -- and 81 us with the third buffer the smaller
// stages leave room for (319 MB of weights + 201 MB of rows per launch = 6.4 TB/s).
template <int KTOT, int MTW>
__global__ void __launch_bounds__(D1_THREADS, 2) k_enc_dense1p(const float *__restrict__ f3, int64_t n_rows_pad,
                                                               const uint4 *__restrict__ wd1x, float *__restrict__ part,
                                                               const caelo_enc_in in) {
    constexpr int BM = D1_BM * MTW, A16 = D1_NS * 4 * BM, NBUF = D1P_NBUF(MTW), NKS = KTOT / D1_SPLIT_OF(KTOT) / D1_BK;
    constexpr int DMA = 4;              // 1 KB pieces a wave copies per stage (8 waves x 4 = 32 >= 26)
    constexpr int PER_STAGE = DMA + MTW;  // loads a thread has in flight per stage: its DMA pieces + its rows
    static_assert(MTW >= 1 && MTW <= 3, "three instances");
    static_assert(NKS % 2 == 0 && NKS >= 6, "stages are unrolled in pairs, the last ones peeled");
    // workgroup -> (row tile, k slice).  De-duplicated launch: only the tiles that hold distinct patches of their frame do work
    // (tiles never straddle frames), and they are numbered FIRST: with one workgroup per CU (LDS) an idle workgroup in the
    // middle of the dispatch order still waits for a whole CU to drain before it can start and exit (96 us for the 62 %
    // of a launch's tiles that are live, against 111 us for all of them, when the idle ones sat where their rows are).
    int64_t row0 = (int64_t)blockIdx.x * BM;
    int split = blockIdx.y;
    if (in.dedup) {
        const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
        int t = lin / D1_SPLIT_OF(KTOT), f = 0;
        split = lin - t * D1_SPLIT_OF(KTOT);
        for (; f < in.n_frames; ++f) {
            const int live = (enc_tables(in, f)->count + BM - 1) / BM;
            if (t < live) break;
            t -= live;
        }
        if (f == in.n_frames) return;
        row0 = (int64_t)f * in.per_frame + (int64_t)t * BM;
    }
    extern __shared__ uint4 d1_lds[];
    uint4 *As = d1_lds;            // [2][split][g][row]
    uint4 *Bs = d1_lds + 2 * A16;  // [NBUF][split][n-tile][lane] (+ 1 KB)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, n = lane & 15;
    const int ks0 = split * NKS;
    const int mg = wave >> 1;          // rows mg * 16 MTW ... of the tile
    const bool odd = (wave & 1) != 0;  // n-tiles 7..12 (6 of them) instead of 0..6
    const int nt0 = odd ? 7 : 0;
    f32x4 acc[MTW][7];
#pragma unroll
    for (int q = 0; q < MTW; ++q)
#pragma unroll
        for (int i = 0; i < 7; ++i) acc[q][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // A fetch: rows a_row (+ 64), 4 consecutive k (16 bytes) from 4 a_kq.  Lane -> (row, a_kq) so that the 16 lanes one LDS store
    // cycle serves (8 rows x the two halves of a 16-byte unit) fall into 16 different bank pairs; (tid >> 3, tid & 7) put four
    // lanes on every bank: 288 conflict cycles per stage, 20 us of the 8-frame launch.  A wave still reads 8 rows x 128 bytes.
    const int a_row = wave * 8 + ((lane >> 1) & 7), a_kq = ((lane >> 4) << 1) | (lane & 1);
    const float *a_src = f3 + (size_t)(row0 + a_row) * KTOT + (size_t)ks0 * D1_BK + a_kq * 4;
    // this wave's middle piece (5 wave + 2) of weight buffer 0 in LDS (the DMA adds 16 x lane itself) and of stage ks0 in memory
    const int pb = wave < 6 ? 4 * wave : 22;   // first of this wave's four pieces
    const uint32_t b_lds = (uint32_t)(uintptr_t)(void __attribute__((address_space(3))) *)(Bs + (pb + 1) * 64);
    const uint4 *b_mid = wd1x + (size_t)ks0 * D1_B16 + (pb + 1) * 64 + lane;
    // LDS byte address of this thread's 8 bytes of A buffer 0, high term ([split][g = a_kq >> 1][row] x 16 B, half a_kq & 1)
    const uint32_t a_lds = (uint32_t)(uintptr_t)(void __attribute__((address_space(3))) *)&As[(a_kq >> 1) * BM + a_row] + (a_kq & 1) * 8;
    f32x4 paA[MTW], paB[MTW];
    d1_h8 af0[MTW][D1_NS], af1[MTW][D1_NS], bf0[4][D1_NS], bf1[3][D1_NS];
    // five CONSECUTIVE 1 KB pieces per wave and stage, addressed from the middle one by the instruction's immediate offset (it
    // moves the global and the LDS address alike): one M0 write and one address register per stage instead of five of each.
    // The eighth wave's fifth piece is piece 39 -- the first KB of the next stage (the weight image is padded by one piece),
    // landing in the buffer's 40th KB, which nobody reads.  Every wave has the same count of loads in flight, which the vmcnt
    // arithmetic below relies on.
#define D1P_FETCH_B(KS, BUF)                                                                                     \
    {                                                                                                            \
        const uint4 *gp_ = b_mid + (size_t)(KS) * D1_B16;                                                        \
        const uint32_t la_ = b_lds + (uint32_t)(BUF) * (D1P_BSTRIDE * 16u);                                      \
        __asm__ volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"                                                       \
                         "global_load_lds_dwordx4 %1, off offset:-1024\n\t"                                       \
                         "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"    \
                         "global_load_lds_dwordx4 %1, off offset:2048" : : "s"(la_), "v"(gp_) : "memory");       \
    }
#define D1P_LOAD_A(P, KS)                                                                                        \
    _Pragma("unroll") for (int r = 0; r < MTW; ++r) {                                                            \
        const float *p_ = a_src + (size_t)(64 * r) * KTOT + (size_t)(KS) * D1_BK;                                \
        __asm__ volatile("global_load_dwordx4 %0, %1, off" : "=v"(P[r]) : "v"(p_) : "memory");                   \
    }
    // wait until at most N loads are in flight; the registers are operands so that no use moves above the wait
#define D1P_WAIT_A(P, N)                                                                                         \
    {                                                                                                            \
        if (MTW == 1) __asm__ volatile("s_waitcnt vmcnt(%1)" : "+v"(P[0]) : "n"(N) : "memory");                  \
        else __asm__ volatile("s_waitcnt vmcnt(%2)" : "+v"(P[0]), "+v"(P[MTW - 1]) : "n"(N) : "memory");         \
    }
    // (the stores are inline asm as well: a compiler-visible LDS store is ordered behind every LDS-DMA in flight -- vmcnt(0))
#define D1P_STORE_A(P, ABUF)                                                                               \
    _Pragma("unroll") for (int r = 0; r < MTW; ++r) {                                                      \
        uint32_t h01_, l01_, h23_, l23_;                                                                   \
        enc_split2h_pk(P[r][0], P[r][1], h01_, l01_);                                                      \
        enc_split2h_pk(P[r][2], P[r][3], h23_, l23_);                                                      \
        const unsigned long long dh_ = ((unsigned long long)h23_ << 32) | h01_, dl_ = ((unsigned long long)l23_ << 32) | l01_; \
        if (r == 0)                                                                                        \
            __asm__ volatile("ds_write_b64 %0, %1 offset:%3\n\tds_write_b64 %0, %2 offset:%4"               \
                             : : "v"(a_lds), "v"(dh_), "v"(dl_), "n"((ABUF) * A16 * 16),                     \
                                 "n"((ABUF) * A16 * 16 + 4 * BM * 16) : "memory");                          \
        else if (r == 1)                                                                                   \
            __asm__ volatile("ds_write_b64 %0, %1 offset:%3\n\tds_write_b64 %0, %2 offset:%4"               \
                             : : "v"(a_lds), "v"(dh_), "v"(dl_), "n"((ABUF) * A16 * 16 + 1024),              \
                                 "n"((ABUF) * A16 * 16 + 4 * BM * 16 + 1024) : "memory");                   \
        else                                                                                               \
            __asm__ volatile("ds_write_b64 %0, %1 offset:%3\n\tds_write_b64 %0, %2 offset:%4"               \
                             : : "v"(a_lds), "v"(dh_), "v"(dl_), "n"((ABUF) * A16 * 16 + 2048),              \
                                 "n"((ABUF) * A16 * 16 + 4 * BM * 16 + 2048) : "memory");                   \
    }
#define D1P_READ_A(AF, ABUF)                                                                               \
    _Pragma("unroll") for (int q = 0; q < MTW; ++q) {                                                      \
        const uint4 *ap_ = &As[(ABUF) * A16 + g * BM + (mg * MTW + q) * 16 + n];                           \
        AF[q][0] = __builtin_bit_cast(d1_h8, ap_[0]);                                                      \
        AF[q][1] = __builtin_bit_cast(d1_h8, ap_[4 * BM]);                                                 \
    }
    // column tiles nt0 .. nt0+3 (group 0) and nt0+4 .. nt0+6 (group 1).  Odd waves own six tiles: their seventh accumulator
    // repeats tile 12 and is dropped at the end -- no branches in the stage; the SIMDs that run the odd waves have the idle
    // slots (36 instead of 42 MFMAs per row tile and stage)
#define D1P_READ_B0(BUF)                                                                                   \
    {                                                                                                      \
        const uint4 *bp_ = &Bs[(BUF) * D1P_BSTRIDE + nt0 * 64 + lane];                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                      \
        _Pragma("unroll") for (int sp = 0; sp < D1_NS; ++sp) bf0[i][sp] = __builtin_bit_cast(d1_h8, bp_[(sp * D1_NT + i) * 64]); \
    }
#define D1P_READ_B1(BUF)                                                                                   \
    {                                                                                                      \
        const uint4 *bp_ = &Bs[(BUF) * D1P_BSTRIDE + lane];                                                \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                    \
            const int t_ = (i == 2 && odd) ? D1_NT - 1 : nt0 + 4 + i;                                      \
            _Pragma("unroll") for (int sp = 0; sp < D1_NS; ++sp) bf1[i][sp] = __builtin_bit_cast(d1_h8, bp_[(sp * D1_NT + t_) * 64]); \
        }                                                                                                  \
    }
    // smallest terms first (AF / B index: 0 high, 1 low); the accumulators alternate
#define D1P_TERM0(AF, SA, SB)                                                                              \
    _Pragma("unroll") for (int q = 0; q < MTW; ++q)                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
        acc[q][i] = D1_MFMA(AF[q][SA], bf0[i][SB], acc[q][i]);
#define D1P_MFMA_G0(AF) D1P_TERM0(AF, 1, 0) D1P_TERM0(AF, 0, 1) D1P_TERM0(AF, 0, 0)
#define D1P_TERM1(AF, SA, SB)                                                                              \
    _Pragma("unroll") for (int q = 0; q < MTW; ++q)                                                        \
    _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                          \
        acc[q][4 + i] = D1_MFMA(AF[q][SA], bf1[i][SB], acc[q][4 + i]);
#define D1P_MFMA_G1(AF) D1P_TERM1(AF, 1, 0) D1P_TERM1(AF, 0, 1) D1P_TERM1(AF, 0, 0)
    // One stage ST whose successor exists.  On entry: bf0 / AFC hold (or are receiving) stage ST's first group and A fragments;
    // in flight, oldest first: weight DMA ST+1 (5), rows ST+1 (registers PN), then YOUNGER more loads of later stages.
    // ISSUE: start DMA ST+NBUF (into the buffer stage ST leaves) and its rows (into PN).  BC / BN: weight buffers of ST / ST+1.
#define D1P_STAGE(ST, AFC, AFN, PN, ABN, BC, BN, YOUNGER, ISSUE)                                                    \
    {                                                                                                               \
        D1P_READ_B1(BC)                                                                                             \
        __builtin_amdgcn_sched_barrier(0); /* the reads stay ahead of the MFMAs that cover them */                  \
        D1P_MFMA_G0(AFC)                                                                                            \
        D1P_WAIT_A(PN, YOUNGER)                                                                                     \
        D1P_STORE_A(PN, ABN)                                                                                        \
        /* rows ST+1 are written, weights ST+1 have landed (older than the rows just waited for), every wave is */  \
        /* done reading stage ST's buffers */                                                                       \
        __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        if (ISSUE) {                                                                                                \
            D1P_FETCH_B((ST) + NBUF, BC)                                                                            \
            D1P_LOAD_A(PN, (ST) + NBUF)                                                                             \
        }                                                                                                           \
        D1P_READ_A(AFN, ABN)                                                                                        \
        D1P_READ_B0(BN)                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        D1P_MFMA_G1(AFC)                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    }
    D1P_FETCH_B(0, 0)
    D1P_LOAD_A(paA, 0)
    D1P_FETCH_B(1, 1)
    D1P_LOAD_A(paB, 1)
    D1P_FETCH_B(2, 2)
    D1P_WAIT_A(paA, PER_STAGE + DMA)  // stage 0: its weights (older) and its rows
    D1P_STORE_A(paA, 0)
    D1P_LOAD_A(paA, 2)
    __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    D1P_READ_A(af0, 0)
    D1P_READ_B0(0)
    int b0 = 0;  // weight buffer of stage st (st % 3)
#pragma unroll 1
    for (int st = 0; st < NKS - 4; st += 2) {
        const int b1 = b0 == 2 ? 0 : b0 + 1, b2 = b1 == 2 ? 0 : b1 + 1;
        D1P_STAGE(st, af0, af1, paB, 1, b0, b1, PER_STAGE, true)
        D1P_STAGE(st + 1, af1, af0, paA, 0, b1, b2, PER_STAGE, true)
        b0 = b2;
    }
    constexpr int c0 = (NKS - 4) % 3, c1 = (NKS - 3) % 3, c2 = (NKS - 2) % 3, c3 = (NKS - 1) % 3;
    D1P_STAGE(NKS - 4, af0, af1, paB, 1, c0, c1, PER_STAGE, true)
    D1P_STAGE(NKS - 3, af1, af0, paA, 0, c1, c2, PER_STAGE, false)
    D1P_STAGE(NKS - 2, af0, af1, paB, 1, c2, c3, 0, false)
    // the last stage: its first group and A fragments are on their way, nothing to prepare
    D1P_READ_B1(c3)
    __builtin_amdgcn_sched_barrier(0);
    D1P_MFMA_G0(af1)
    D1P_MFMA_G1(af1)
    // C rows 4g + r of each tile
#pragma unroll
    for (int q = 0; q < MTW; ++q)
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            if (i == 6 && odd) continue;
            float *dst = part + ((size_t)split * n_rows_pad + row0 + (mg * MTW + q) * 16 + 4 * g) * DENSE_NP + (nt0 + i) * 16 + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(size_t)r * DENSE_NP] = acc[q][i][r];
        }
}

// Per-DEVICE caches of things the runtime is asked once: function attributes and occupancy answers belong to the device that is
// current when they are set / asked (process-wide statics would serve a second GPU the first one's answers).
#include <mutex>
#define CAELO_MAX_DEVICES 64
template <typename T, typename F>
static T per_device_once(int device, T (&slot)[CAELO_MAX_DEVICES], bool (&have)[CAELO_MAX_DEVICES], std::mutex &mu, F make) {
    const int d = device >= 0 && device < CAELO_MAX_DEVICES ? device : 0;
    std::lock_guard<std::mutex> lock(mu);
    if (!have[d]) { slot[d] = make(); have[d] = true; }
    return slot[d];
}

template <int KTOT>
static int dense1_launch(int device, const float *f3, int64_t np, const void *wd1x, float *part, const caelo_enc_in &in, hipStream_t s) {
    // once per device (under a lock: the pipeline's encoder thread and the caller may race here)
    static hipError_t attr_slot[CAELO_MAX_DEVICES];
    static bool attr_have[CAELO_MAX_DEVICES];
    static std::mutex attr_mu;
    const hipError_t attr = per_device_once(device, attr_slot, attr_have, attr_mu, [] {
        hipError_t e = hipFuncSetAttribute((const void *)k_enc_dense1p<KTOT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, D1P_LDS_BYTES(1));
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)k_enc_dense1p<KTOT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, D1P_LDS_BYTES(2));
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)k_enc_dense1p<KTOT, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, D1P_LDS_BYTES(3));
        return e;
    });
    CAELO_HIP(attr);
    // CAELO_D1_WIDE_FROM: launch size (rows) from which the 128-row instance is used -- a tuning threshold, the partial sums of the two
    // instances are bit-identical (tests/test_gpu_parity.py::test_dense1_tile_sizes_are_bit_identical)
    static const int64_t wide_from = getenv("CAELO_D1_WIDE_FROM") ? atoll(getenv("CAELO_D1_WIDE_FROM")) : 4 * 3072;  // rows
    // 192-row tiles (round 6) for launches of EVERY patch (no de-duplication) that 128-row tiles would take two rounds over (more than
    // 256 x 64 rows): one and a half times the MFMAs per weight fragment, two thirds of the weight stream, one round -- 90 -> 68 us per
    // 24 576 rows; 12 288 rows (one round either way) 45 -> 56 us, which is why de-duplicated launches (~13 k live rows of a batch, a
    // number the host does not know) stay on 128-row tiles: 19.7 against 19.5 k frames/s (profiles/r06_dense1_tile192.txt).  Same partial sums.
    static const int64_t tile3_from = getenv("CAELO_D1_TILE3_FROM") ? atoll(getenv("CAELO_D1_TILE3_FROM")) : 256 * 64 + 1;  // rows
    if (KTOT == 2048 && !in.dedup && np % 192 == 0 && np >= tile3_from && np >= wide_from) {
        dim3 gw((unsigned)(np / 192), D1_SPLIT_OF(KTOT));
        k_enc_dense1p<KTOT, 3><<<gw, D1_THREADS, D1P_LDS_BYTES(3), s>>>(f3, np, (const uint4 *)wd1x, part, in);
    } else
    if (np % 128 == 0 && (!in.dedup || in.per_frame % 128 == 0) && (np >= wide_from || KTOT > 2048)) {  // (tiles never straddle frames)
        // 128-row tiles once the launch fills the chip with them, and always for the long-K instance (8 k slices per row tile):
        // half the weight stream; same partial sums
        dim3 gw((unsigned)(np / 128), D1_SPLIT_OF(KTOT));
        k_enc_dense1p<KTOT, 2><<<gw, D1_THREADS, D1P_LDS_BYTES(2), s>>>(f3, np, (const uint4 *)wd1x, part, in);
    } else {
        dim3 gd((unsigned)(np / 64), D1_SPLIT_OF(KTOT));
        k_enc_dense1p<KTOT, 1><<<gd, D1_THREADS, D1P_LDS_BYTES(1), s>>>(f3, np, (const uint4 *)wd1x, part, in);
    }
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

// ------------------------------------------------------------------------------------------------
// k_enc_head_mfma (round 3): the same arithmetic up to the hidden layer (bias + the k slices in slice order, tanh), then Dense(20) of
// SIXTEEN patches per workgroup on v_mfma_f32_16x16x4_f32 (an exact fmaf chain over k ascending) instead of one patch per
// wavefront with a 22-shuffle reduce-scatter behind a serial chain per patch.  The four wavefronts of a workgroup split the 13
// groups of 16 hidden columns (j = wave, wave + 4, ...); everything a wavefront needs after the row lookup is requested at once --
// its slices of the partial sums, the bias, its fragments of the weights (enc_wd2q, pre-arranged) -- so a tile is two memory
// round trips deep; the four partial products meet in LDS and are added in wavefront order.
//   A[m][k]: lane (m = lane & 15, g = lane >> 4) holds hidden columns 16 j + 4 g .. + 3 of patch m; MFMA (j, i) contracts
//   k = 16 j + 4 g + i over g.  B[k][n] = wd2 zero-padded to [208][32].
//   C[4 g + i][n]: lane (g, n) holds outputs n (n-tile 0) and 16 + n (n-tile 1: four live columns) of patches 4 g .. 4 g + 3.
// The sum over the 200 hidden units is four chains in ascending k here and a tree there: descriptors differ from k_enc_head's in
// the last bits (2e-7), the oracle's tolerance is 1e-4 (CAELO_ENC_HEAD=wave selects the old kernel).
#define HM_JW ((D1_NT + 3) / 4)   // hidden-column groups per wavefront (4, the last wavefront's fourth is empty)
// host: wd2 [200][20] -> [j 13][n-tile 2][lane 64] x float4 (i = 0 .. 3): w[16 j + 4 g + i][16 nt + n], zero beyond 200 / 20
static void head_weight_fragments(const float *wd2, float *out) {
    for (int j = 0; j < D1_NT; ++j)
        for (int nt = 0; nt < 2; ++nt)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i) {
                    const int k = 16 * j + 4 * (lane >> 4) + i, col = 16 * nt + (lane & 15);
                    out[(((size_t)j * 2 + nt) * 64 + lane) * 4 + i] = (k < DENSE_N && col < 20) ? wd2[k * 20 + col] : 0.0f;
                }
}

template <int SPLIT>
__global__ void __launch_bounds__(256) k_enc_head_mfma(const float *__restrict__ part, int64_t n_patches, int64_t n_rows_pad,
                                                       const float *__restrict__ bd1p, const float4 *__restrict__ wd2q,
                                                       const float *__restrict__ bd2, int group, caelo_enc_out outs,
                                                       int out_stride, const caelo_enc_in in, int32_t *faults) {
    __shared__ f32x4 s_c[3][2][64];   // partial products of wavefronts 1 .. 3
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, g = lane >> 4;
    const unsigned n_all = (unsigned)n_patches, per_in = (unsigned)in.per_frame, per_out = (unsigned)outs.per_frame;
    const unsigned tile = blockIdx.x;
    const unsigned p_raw = tile * 16u + (unsigned)m;
    const unsigned p = p_raw < n_all ? p_raw : n_all - 1u;   // (a lane past the end repeats the last patch; its outputs are not stored)
    // ---- requests that do not depend on the patch: this wavefront's weight fragments and bias
    float4 wq[HM_JW][2], b1[HM_JW];
#pragma unroll
    for (int c = 0; c < HM_JW; ++c) {
        const int j = wave + 4 * c;
        if (j < D1_NT) {
            wq[c][0] = wd2q[((size_t)j * 2 + 0) * 64 + lane];
            wq[c][1] = wd2q[((size_t)j * 2 + 1) * 64 + lane];
            b1[c] = *(const float4 *)(bd1p + 16 * j + 4 * g);
        }
    }
    // ---- the row that holds the patch's result: its representative's with de-duplication
    unsigned row = p;
    if (in.dedup) {
        const unsigned f_ = p / per_in;
        int r_ = enc_tables(in, (int)f_)->slot_of[p - f_ * per_in];        // a row of the whole launch set ...
        if (r_ < 0) {                                                       // ... or -(representative + 1): its row
            const unsigned g_ = (unsigned)(-r_ - 1), fr_ = g_ / per_in;
            r_ = enc_tables(in, (int)fr_)->slot_of[g_ - fr_ * per_in];
        }
        row = (unsigned)r_;
    }
    const float *src = part + (size_t)row * DENSE_NP + 4 * g;
    float4 v[HM_JW][SPLIT];
#pragma unroll
    for (int c = 0; c < HM_JW; ++c)
#pragma unroll
        for (int sp = 0; sp < SPLIT; ++sp)
            if (wave + 4 * c < D1_NT) v[c][sp] = *(const float4 *)(src + (size_t)sp * n_rows_pad * DENSE_NP + 16 * (wave + 4 * c));
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < HM_JW; ++c) {
        if (wave + 4 * c < D1_NT) {
            float4 sum = b1[c];
#pragma unroll
            for (int sp = 0; sp < SPLIT; ++sp) { sum.x += v[c][sp].x; sum.y += v[c][sp].y; sum.z += v[c][sp].z; sum.w += v[c][sp].w; }
            const float hv[4] = {enc_tanh(sum.x), enc_tanh(sum.y), enc_tanh(sum.z), enc_tanh(sum.w)};
            const float w0[4] = {wq[c][0].x, wq[c][0].y, wq[c][0].z, wq[c][0].w}, w1[4] = {wq[c][1].x, wq[c][1].y, wq[c][1].z, wq[c][1].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[i], w0[i], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[i], w1[i], acc1, 0, 0, 0);
            }
        }
    }
    if (wave > 0) {
        s_c[wave - 1][0][lane] = acc0;
        s_c[wave - 1][1][lane] = acc1;
    }
    __syncthreads();
    if (wave > 0) return;
    // ---- wavefront 0: the four partial products in wavefront order, bias, tanh; C row 4 g + i = patch tile * 16 + 4 g + i
#pragma unroll
    for (int w = 0; w < 3; ++w) {
        acc0 += s_c[w][0][lane];
        acc1 += s_c[w][1][lane];
    }
    const float bo0 = bd2[m], bo1 = m < 4 ? bd2[16 + m] : 0.0f;   // output bias of this lane's two C columns
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned pp = tile * 16u + 4u * (unsigned)g + (unsigned)i;
        if (pp < n_all) {
            // patches of several frames in one launch: frame f = p / per_frame writes into its own rows
            const unsigned f = pp / per_out, q = pp - f * per_out, kp = q / (unsigned)group;
            float *dst = outs.base[f] + (size_t)kp * out_stride + (size_t)(q - kp * (unsigned)group) * 20;
            // the encoder's own invariant (see k_enc_head): a descriptor is a tanh -- finite and within [-1, 1]; counted per value
            const float d0 = enc_tanh(bo0 + acc0[i]);
            dst[m] = d0;
            if (!(fabsf(d0) <= 1.0f)) atomicAdd(faults, 1);
            if (m < 4) {
                const float d1 = enc_tanh(bo1 + acc1[i]);
                dst[16 + m] = d1;
                if (!(fabsf(d1) <= 1.0f)) atomicAdd(faults, 1);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host entry
// ------------------------------------------------------------------------------------------------
static inline int64_t pad64(int64_t n) { return (n + D1_BM - 1) / D1_BM * D1_BM; }  // rows padded to whole dense-1 tiles

// per row: counts (4 quarters x 4 B) + cells (512 x 2 B) + values (512 x 8 x 4 B) of the non-background cells after conv1 + pool1

CAELO_API int64_t caelo_encode_ws_bytes(int64_t n_patches) {
    const int64_t np = pad64(n_patches);
    return CAELO_ENC_WS_HEADER + (np * 1024 + np * 2048 + (int64_t)D1_SPLIT * np * DENSE_NP) * (int64_t)sizeof(float);
}

CAELO_API int caelo_encode_ws_layout(int64_t n_patches, int64_t out[6]) {
    CAELO_REQUIRE(out && n_patches > 0, "caelo_encode_ws_layout: bad argument");
    const int64_t np = pad64(n_patches);
    out[0] = CAELO_ENC_WS_HEADER;
    out[1] = out[0] + np * 1024 * (int64_t)sizeof(float);
    out[2] = out[1] + np * 2048 * (int64_t)sizeof(float);
    out[3] = np;
    out[4] = D1_SPLIT_OF(DENSE_K);
    out[5] = D1_SPLIT;
    return CAELO_OK;
}

int encode_impl(caelo_ctx *c, const uint64_t *bits, int64_t n_patches, int group, float *out, int out_stride,
                       void *ws, hipStream_t s, hipEvent_t *ev /* 5 events or null */) {
    CAELO_REQUIRE(out, "null argument");
    caelo_enc_out outs;
    outs.base[0] = out;
    outs.per_frame = n_patches > 0 ? n_patches : 1;
    return encode_batch_impl(c, bits, n_patches, group, outs, out_stride, ws, s, ev);
}

int encode_batch_impl(caelo_ctx *c, const uint64_t *bits, int64_t n_patches, int group, const caelo_enc_out &outs,
                      int out_stride, void *ws, hipStream_t s, hipEvent_t *ev /* 5 events or null */, const caelo_enc_in *in) {
    CAELO_REQUIRE(c && bits && ws, "null argument");
    // plain launch: every patch, contiguous
    const caelo_enc_in ein = in ? *in : caelo_enc_in{(const unsigned long long *)bits, 0, (int32_t)(n_patches < 0x7FFFFFFF ? n_patches : 0), 1, 0, 0};
    CAELO_REQUIRE(!ein.dedup || (ein.n_frames >= 1 && ein.n_frames <= CAELO_ENC_MAX_FRAMES && ein.per_frame % (3 * D1_BM) == 0 &&
                                 (int64_t)ein.n_frames * ein.per_frame == n_patches),
                  "bad de-duplicated launch");
    CAELO_REQUIRE(outs.per_frame > 0 && (n_patches + outs.per_frame - 1) / outs.per_frame <= CAELO_ENC_MAX_FRAMES, "bad frame table");
    // (the de-duplication tables address representatives by row of the launch set = frame * 3072 + position: dedup.hip)
    CAELO_REQUIRE(!ein.dedup || ein.per_frame == CAELO_FRAME_PATCHES, "de-duplicated launches hold whole frames of 3072 patches");
    CAELO_REQUIRE(c->has_enc, "encoder weights not set (caelo_set_encoder_weights)");
    CAELO_REQUIRE(n_patches > 0 && n_patches < 0x7FFFFFFF && group >= 1 && out_stride >= group * 20, "bad shape");
    const int64_t np = pad64(n_patches);
    // ws = [header (CAELO_ENC_WS_HEADER bytes): work counter, zero between calls (the owner zero-fills ws once, conv3 resets it); bytes 3072..4095
    // = MFMA instructions the last profiled stage-1 launch executed, eight partial counts] | P2 | F3 | dense-1 partial sums | conv-1 cell lists
    int *work_counter = (int *)ws;
    unsigned long long *mfma_count = (unsigned long long *)((char *)ws + CAELO_ENC_WS_MFMA);   // eight counters, a 128-byte line each
    float *p2 = (float *)((char *)ws + CAELO_ENC_WS_HEADER);
    float *f3 = p2 + np * 1024;
    float *part = f3 + np * 2048;
    if (np > n_patches)  // rows of the last 64-row tile that no patch writes
        CAELO_HIP(hipMemsetAsync(f3 + n_patches * 2048, 0, (size_t)(np - n_patches) * 2048 * sizeof(float), s));
    // persistent grid = exactly the resident workgroup slots (weights stay in registers across patches,
    // no second partially filled round)
    // Stage 1 = k_enc_stage1x (enc_stage1x.inc: conv1 on the matrix cores, f16 x 2 products, two 256-register workgroups per CU).
    // The exact-f32 k_enc_stage1 of round 2 stays as the PRECISION REFERENCE, chosen per context by caelo_set_encoder_reference --
    // never by the environment: the library reads no variable that changes arithmetic.  (The kernels rounds 1-3 measured and lost
    // with -- one wavefront per patch, conv1 / conv2 as two kernels, the two-barrier Dense(200), the one-patch-per-wavefront head, the
    // VALU response layer, the all-f64 match as a default -- are gone from the library; DESIGN.md 4 keeps their numbers.)
    const bool stage1x = !c->enc_reference;
    static int slots1x_slot[CAELO_MAX_DEVICES];
    static bool slots1x_have[CAELO_MAX_DEVICES];
    static std::mutex slots1x_mu;
    const int slots1x = per_device_once(c->device, slots1x_slot, slots1x_have, slots1x_mu, [&] {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_enc_stage1x<false>, 256, 0) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || per_cu * cus <= 0)
            return 512;
        return per_cu * cus;
    });
    int *xcd_counters = (int *)((char *)ws + 1024);  // 8 x one 128-byte line
    static int slots1_slot[CAELO_MAX_DEVICES];
    static bool slots1_have[CAELO_MAX_DEVICES];
    static std::mutex slots1_mu;
    const int slots1 = per_device_once(c->device, slots1_slot, slots1_have, slots1_mu, [&] {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_enc_stage1, 256, 0) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || per_cu * cus <= 0)
            return 768;  // 3 workgroups on each of MI355X's 256 CUs
        return per_cu * cus;
    });
    // Inside the frame pipeline the persistent grids leave a fifth (stage 1) / a quarter (conv3) of their slots free: stage 1 and
    // conv3 otherwise own every register file for their whole run and the other streams' kernels only get CUs between them
    // (round 1: +3 % frames/s; with the round-2 executor the setting is worth about 1 %, caelo_enc_in::yield bits)
    const int64_t cap1 = (ein.yield & 1) ? (int64_t)slots1 * 4 / 5 : slots1;
    const unsigned g1 = (unsigned)(n_patches < cap1 ? n_patches : cap1);
    if (ev) CAELO_HIP(hipEventRecord(ev[0], s));
    const int order_group = (n_patches % group == 0) ? group : 1;
    if (stage1x) {
        // CAELO_S1X_SLOTS: grid size by hand (measurement: 128 / 256 / 384 / 512 workgroups take 692 / 376 / 281 / 233 us per 24 576
        // patches -- a workgroup alone on its CU needs 3.9 us per patch, two sharing one 4.85 us each: latency bound)
        static const int slots_env = getenv("CAELO_S1X_SLOTS") ? atoi(getenv("CAELO_S1X_SLOTS")) : 0;
        // inside the pipeline (yield bit 0): two of the three workgroups a CU can hold -- the third's registers and LDS go to the front
        // and pair kernels of the other streams (frames/s at 120 batches against the grid: 448 / 512 / 576 / 640 / 704 / 768 workgroups
        // = 20.0 / 20.4 / 20.2 / 19.6 / 19.2 / 17.8 k, profiles/r04_pipe_sweep.txt); alone, all three (207 against 248 us)
        const int64_t capx = slots_env > 0 ? slots_env : ((ein.yield & 1) ? (int64_t)slots1x * 2 / 3 : slots1x);
        const unsigned gx = (unsigned)(n_patches < capx ? n_patches : capx);
        if (ev) {   // profiling calls count the MFMAs the kernel executes (bench.py's roofline); same code otherwise
            CAELO_HIP(hipMemsetAsync(mfma_count, 0, 1024, s));
            CAELO_HIP(hipEventRecord(ev[0], s));
            k_enc_stage1x<true><<<gx, 256, 0, s>>>(ein, n_patches, order_group, work_counter, (const uint4 *)c->enc_w1f, c->enc_b1,
                                                   (const uint4 *)c->enc_w2x, c->enc_c0, p2, mfma_count);
        } else
            k_enc_stage1x<false><<<gx, 256, 0, s>>>(ein, n_patches, order_group, work_counter, (const uint4 *)c->enc_w1f, c->enc_b1,
                                                    (const uint4 *)c->enc_w2x, c->enc_c0, p2, mfma_count);
        CAELO_LAUNCH_CHECK();
    } else {
        if (ev) CAELO_HIP(hipMemsetAsync(mfma_count, 0, 1024, s));   // (the f32 kernel has no register left to count)
        k_enc_stage1<<<g1, 256, 0, s>>>(ein, n_patches, order_group, work_counter, c->enc_w1, c->enc_b1, c->enc_w2, c->enc_c0, p2);
        CAELO_LAUNCH_CHECK();
    }
    if (ev) CAELO_HIP(hipEventRecord(ev[1], s));
    const int64_t pairs = (n_patches + 1) / 2;
    const int64_t cap3 = ((ein.yield & 2) ? 256 : ((ein.yield & 4) ? 384 : 512)) * C3X_WGS / 2;
    const unsigned g3 = (unsigned)(pairs < cap3 ? pairs : cap3);  // persistent: two 4-wave workgroups per CU
    k_enc_conv3<<<g3, 256, 0, s>>>(p2, n_patches, ein, (const uint4 *)c->enc_w3x, c->enc_b3, f3, work_counter, xcd_counters);
    CAELO_LAUNCH_CHECK();
    if (ev) CAELO_HIP(hipEventRecord(ev[2], s));
    {
        const int rc = dense1_launch<DENSE_K>(c->device, f3, np, c->enc_wd1x, part, ein, s);
        if (rc) return rc;
    }
    if (ev) CAELO_HIP(hipEventRecord(ev[3], s));
    k_enc_head_mfma<D1_SPLIT_OF(DENSE_K)><<<(unsigned)((n_patches + 15) / 16), 256, 0, s>>>(part, n_patches, np, c->enc_bd1, (const float4 *)c->enc_wd2q,
                                                                                       c->enc_bd2, group, outs, out_stride, ein, c->faults);
    CAELO_LAUNCH_CHECK();
    if (ev) CAELO_HIP(hipEventRecord(ev[4], s));
    return CAELO_OK;
}

// Dense(200) + Dense(20) of the 32^3 stress case (config5.hip): the same two kernels over K = 16384
int64_t enc_dense_pad(int64_t n) { return pad64(n); }
int64_t enc_dense32_part_bytes(int64_t np) { return (int64_t)D1_SPLIT * np * DENSE_NP * (int64_t)sizeof(float); }
int enc_dense32_head_launch(caelo_ctx *c, const float *f3, int64_t n_patches, int64_t np, float *part, int group, float *out,
                            int out_stride, hipStream_t s) {
    const caelo_enc_in plain = {nullptr, 0, (int32_t)n_patches, 1, 0, 0};
    {
        const int rc = dense1_launch<16384>(c->device, f3, np, c->enc32_wd1x, part, plain, s);
        if (rc) return rc;
    }
    caelo_enc_out outs = {};
    outs.base[0] = out;
    outs.per_frame = n_patches;
    k_enc_head_mfma<D1_SPLIT_OF(16384)><<<(unsigned)((n_patches + 15) / 16), 256, 0, s>>>(part, n_patches, np, c->enc32_bd1, (const float4 *)c->enc_wd2q,
                                                                                     c->enc_bd2, group, outs, out_stride, plain, c->faults);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

CAELO_API int caelo_encode(caelo_ctx *c, const uint64_t *bits, int64_t n_patches, int group, float *out,
                           int out_stride, void *ws, void *stream) {
    return encode_impl(c, bits, n_patches, group, out, out_stride, ws, caelo_stream(stream), nullptr);
}

// Same launches as caelo_encode with a HIP event between kernels on the launch stream; synchronises and
// returns the four kernel durations in ms (stage1, conv3, dense1, head).  Measurement aid for bench.py.
CAELO_API int caelo_encode_profile(caelo_ctx *c, const uint64_t *bits, int64_t n_patches, int group, float *out,
                                   int out_stride, void *ws, void *stream, float *ms_host) {
    CAELO_REQUIRE(ms_host != nullptr, "null argument");
    hipStream_t s = caelo_stream(stream);
    hipEvent_t ev[5];
    for (int i = 0; i < 5; ++i) CAELO_HIP(hipEventCreate(&ev[i]));
    int rc = encode_impl(c, bits, n_patches, group, out, out_stride, ws, s, ev);
    if (rc == CAELO_OK) {
        CAELO_HIP(hipEventSynchronize(ev[4]));
        for (int i = 0; i < 4; ++i) CAELO_HIP(hipEventElapsedTime(&ms_host[i], ev[i], ev[i + 1]));
        unsigned long long rows = 0, part[128];  // MFMA instructions executed, counted by the kernel itself (eight partial counts, a line each)
        CAELO_HIP(hipMemcpy(part, (char *)ws + CAELO_ENC_WS_MFMA, sizeof(part), hipMemcpyDeviceToHost));
        for (int i = 0; i < 8; ++i) rows += part[16 * i];
        // the f32 kernel counts tap rows (6 v_mfma_f32_16x16x4_f32 each), k_enc_stage1x its v_mfma_f32_16x16x32_f16 instructions
        const bool f32_kernel = c->enc_reference;   // (it does not count: 0 MFMAs reported)
        ms_host[4] = (float)((double)rows * (f32_kernel ? 6.0 : 1.0) / 1e6);
        ms_host[5] = f32_kernel ? 2.0f * 16 * 16 * 4 : 2.0f * 16 * 16 * 32;   // FLOPs of one counted instruction
    }
    for (int i = 0; i < 5; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

// The exact-f32 stage 1 of round 2 (k_enc_stage1: f32-input MFMAs for conv2, conv1 on the VALU) as this context's stage 1: the
// precision reference the f16 x 2 kernel is measured against (tests, tools/enc_layer_errors.py).  Slower (346 vs 224 us per
// 24 576 patches); everything behind stage 1 is unchanged.
CAELO_API int caelo_set_encoder_reference(caelo_ctx *c, int on) {
    CAELO_REQUIRE(c, "null argument");
    c->enc_reference = on != 0;
    return CAELO_OK;
}
