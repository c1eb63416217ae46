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
    float *enc_w3;   // [27][16][32]
    float *enc_b3;   // [32]
    float *enc_wd1;  // [2048][208] (N padded 200 -> 208 with zeros)
    float *enc_bd1;  // [208]
    float *enc_wd2;  // [200][20]
    float *enc_bd2;  // [20]
    bool has_enc;
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

struct caelo_voxmap {
    int64_t max_points;
    caelo_brick_table brick[3];
    // voxel-level first-touch tables (value = smallest inserting point index)
    unsigned long long *vkeys[3];
    int32_t *vfirst[3];
    uint32_t vmask[3];
    int32_t *counts;  // [4] device: unique voxels per scale, [3] = spare
    // export scratch
    void *scratch;
    int64_t scratch_bytes;
};

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
