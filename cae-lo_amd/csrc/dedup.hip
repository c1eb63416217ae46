// dedup.hip -- exact de-duplication of a frame's 3 x 1024 voxel patches before the encoder.
//
// GetFeaturesFromPatches (Match.py:130-135) encodes every patch of GetPatchesList (Voxel.py:177-216) on its own, and
// the encoder is a pure function of the patch bits.  Key points come in clusters (GetKeyPtsByAE keeps the top-1024
// response pixels, SphericalRing.py:113-291: neighbouring pixels along an edge), so at the 16 cm and 64 cm scales
// many of them share the key voxel and therefore the patch, and at the 2 cm scale most patches are the key voxel
// plus one or two neighbours in a handful of arrangements: a 64-beam frame has ~1.9 k distinct patches among its
// 3072 (283 / 950 / 622 at the three scales).  Equal bits => equal descriptor, bit for bit (the encoder kernels do
// not depend on a patch's position in the launch: tests/test_gpu_parity.py), so the encoder runs once per distinct
// patch and k_enc_head hands the result to every key point that owns a copy.
//
// Exactness: patches are grouped by a 40-bit hash, then every patch is compared word by word with the
// representative of its group (the smallest patch index with that hash); a patch that differs (hash collision)
// stays its own representative.  CAELO_DEDUP_HASH_BITS=<n> shrinks the hash to n bits to exercise that path.
//
// Output (caelo_dedup_tables, behind the frame's bit-packed patches): count, list[count] = the representatives,
// coarsest scale first (the encoder's heavy-first order), slot_of[patch] = position of its representative in list.
#include <stdlib.h>

#include "caelo_internal.h"

#define DD_SLOTS 8192  // >= 2.6 x the 3072 patches of a frame
#define DD_EMPTY 0xFFFFFFFFFFFFFFFFull

struct DedupScratch {
    unsigned long long table[DD_SLOTS];  // (hash40 << 24) | smallest patch index, DD_EMPTY = free (cleared per frame)
    int32_t pslot[CAELO_FRAME_PATCHES];
    int32_t rep[CAELO_FRAME_PATCHES];
};

int64_t dedup_scratch_bytes() { return (int64_t)sizeof(DedupScratch); }
void dedup_clear_item(void *scratch, caelo_clear_list &list) {
    list.item[list.n++] = {scratch, sizeof(unsigned long long) * DD_SLOTS, 0xFFFFFFFFu};
}

__device__ inline unsigned long long dd_mix(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

// one wavefront per patch: hash the 64 words, claim / join the hash's table entry
__global__ void __launch_bounds__(256) k_dd_insert(const unsigned long long *__restrict__ bits, DedupScratch *S,
                                                   unsigned long long hash_mask) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned long long h = dd_mix(bits[(size_t)p * 64 + lane] + 0x9E3779B97F4A7C15ull * (unsigned)(lane + 1));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);  // order independent across lanes, position dependent per word
    if (lane != 0) return;
    h = dd_mix(h) & hash_mask & 0xFFFFFFFFFEull;  // 40 bits, never all ones
    const unsigned long long mine = (h << 24) | (unsigned)p;
    uint32_t slot = (uint32_t)(dd_mix(h) & (DD_SLOTS - 1));
    for (;;) {
        unsigned long long cur = S->table[slot];
        if (cur == DD_EMPTY) {
            cur = atomicCAS(&S->table[slot], DD_EMPTY, mine);
            if (cur == DD_EMPTY) break;
        }
        if ((cur >> 24) == h) {
            // hundreds of patches share a popular pattern: only a smaller index than the one seen needs the atomic (a
            // stale larger value only costs an atomic that changes nothing)
            if (mine < cur) atomicMin(&S->table[slot], mine);
            break;
        }
        slot = (slot + 1) & (DD_SLOTS - 1);
    }
    S->pslot[p] = (int32_t)slot;
}

// one wavefront per patch: word-by-word comparison with the group's representative
__global__ void __launch_bounds__(256) k_dd_verify(const unsigned long long *__restrict__ bits, DedupScratch *S) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r = (int)(S->table[S->pslot[p]] & 0xFFFFFFull);
    bool same = true;
    if (r != p) same = __all(bits[(size_t)p * 64 + lane] == bits[(size_t)r * 64 + lane]) != 0;
    if (lane == 0) S->rep[p] = same ? r : p;
}

// one workgroup: the representatives in encoder order (scale 2, 1, 0; key point index within a scale)
__global__ void __launch_bounds__(1024) k_dd_scan(const DedupScratch *S, caelo_dedup_tables *T, int identity) {
    __shared__ int wsum[16];
    __shared__ int pos_of[CAELO_FRAME_PATCHES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int flag[3], p[3];
    int mine = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int o = tid * 3 + j;  // order index: scale 2 first
        p[j] = (o & 1023) * 3 + (2 - (o >> 10));
        flag[j] = identity || S->rep[p[j]] == p[j];
        mine += flag[j];
    }
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) base += wsum[w];
        total += wsum[w];
    }
    int pos = base + incl - mine;
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (flag[j]) {
            T->list[pos] = p[j];
            pos_of[p[j]] = pos;
            ++pos;
        }
    if (tid == 0) T->count = total;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 3; ++j) T->slot_of[p[j]] = pos_of[identity ? p[j] : S->rep[p[j]]];
}

// bits: the frame's [3072][64] u64 patches, followed by its caelo_dedup_tables (caelo_frame_tables)
int dedup_launch(uint64_t *bits, void *scratch, bool enabled, hipStream_t s) {
    static const unsigned long long mask = [] {
        const char *e = getenv("CAELO_DEDUP_HASH_BITS");
        const int nb = e ? atoi(e) : 40;
        return nb >= 40 || nb < 1 ? 0xFFFFFFFFFFull : ((1ull << nb) - 1ull) << 1;
    }();
    static const bool off = [] {
        const char *d = getenv("CAELO_NO_DEDUP");
        return d && atoi(d);
    }();
    DedupScratch *S = (DedupScratch *)scratch;
    caelo_dedup_tables *T = caelo_frame_tables(bits);
    const bool on = enabled && !off;
    if (on) {
        k_dd_insert<<<CAELO_FRAME_PATCHES / 4, 256, 0, s>>>((const unsigned long long *)bits, S, mask);
        CAELO_LAUNCH_CHECK();
        k_dd_verify<<<CAELO_FRAME_PATCHES / 4, 256, 0, s>>>((const unsigned long long *)bits, S);
        CAELO_LAUNCH_CHECK();
    }
    k_dd_scan<<<1, 1024, 0, s>>>(S, T, on ? 0 : 1);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}
