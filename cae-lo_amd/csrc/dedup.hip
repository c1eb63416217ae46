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
// Exactness: patches are grouped by a 40-bit hash (computed and entered into the table by k_patches, which has the
// words in registers), then every patch is compared word by word with the
// representative of its group (the smallest patch index with that hash); a patch that differs (hash collision)
// stays its own representative.  CAELO_DEDUP_HASH_BITS=<n> shrinks the hash to n bits to exercise that path.
//
// Output (caelo_dedup_tables, behind the frame's bit-packed patches): count, list[count] = the representatives,
// coarsest scale first (the encoder's heavy-first order), slot_of[patch] = position of its representative in list.
#include <stdlib.h>

#include "caelo_internal.h"

int64_t dedup_scratch_bytes() { return (int64_t)sizeof(DedupScratch); }
void dedup_clear_item(void *scratch, int n_frames, caelo_clear_list &list) {
    list.item[list.n++] = {scratch, sizeof(unsigned long long) * ((size_t)dedup_slot_mask(n_frames) + 1), 0xFFFFFFFFu};
}

// (the per-patch hash + table insert runs at the end of k_patches, voxel.hip: caelo_dedup_insert in caelo_internal.h)

// one wavefront per patch: word-by-word comparison with the group's representative
// the candidate representative of a patch = the smallest (frame, patch) with its hash; equal only if the 512 bytes are
__global__ void __launch_bounds__(256) k_dd_verify(const caelo_frame_set fs) {
    const unsigned long long *__restrict__ bits = fs.f[blockIdx.z].bits;
    DedupScratch *S = fs.f[blockIdx.z].dd;
    const DedupScratch *T = fs.f[0].dd;
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), gp = blockIdx.z * CAELO_FRAME_PATCHES + p;
    const int r = (int)(T->table[S->pslot[p]] & 0xFFFFFFull);
    bool same = true;
    if (r != gp) {
        const int fr = r / CAELO_FRAME_PATCHES, rl = r - fr * CAELO_FRAME_PATCHES;
        same = __all(bits[(size_t)p * 64 + lane] == fs.f[fr].bits[(size_t)rl * 64 + lane]) != 0;
    }
    if (lane == 0) S->rep[p] = same ? r : gp;
}

// per frame: the list of its distinct patches (those that represent themselves), coarsest scale first, and their rows
__global__ void __launch_bounds__(1024) k_dd_scan(const caelo_frame_set fs, int identity) {
    const DedupScratch *S = fs.f[blockIdx.z].dd;
    caelo_dedup_tables *T = caelo_frame_tables((const uint64_t *)fs.f[blockIdx.z].bits);
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = blockIdx.z * CAELO_FRAME_PATCHES;
    int flag[3], p[3];
    int mine = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int o = tid * 3 + j;  // order index: scale 2 first
        p[j] = (o & 1023) * 3 + (2 - (o >> 10));
        flag[j] = identity || S->rep[p[j]] == base + p[j];
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
    int off = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) off += wsum[w];
        total += wsum[w];
    }
    int pos = off + incl - mine;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (flag[j]) {
            T->list[pos] = p[j];
            T->slot_of[p[j]] = base + pos;   // its own row
            ++pos;
        } else {
            // a copy: -(representative + 1), the representative being (frame * 3072 + patch) of any frame of the set; the one
            // reader (k_enc_head) follows it to that patch's own row -- a second lookup for the copies instead of a kernel
            // (k_dd_link, 5 us on the front stream's serial chain) that resolved it for them
            T->slot_of[p[j]] = -(S->rep[p[j]] + 1);
        }
    }
    if (tid == 0) T->count = total;
}

unsigned long long dedup_hash_mask() {
    static const unsigned long long mask = [] {
        const char *e = getenv("CAELO_DEDUP_HASH_BITS");
        const int nb = e ? atoi(e) : 40;
        return nb >= 40 || nb < 1 ? 0xFFFFFFFFFFull : ((1ull << nb) - 1ull) << 1;
    }();
    return mask;
}
bool dedup_enabled(int mode) {
    static const bool off = [] {
        const char *d = getenv("CAELO_NO_DEDUP");
        return d && atoi(d);
    }();
    return !off && !(mode & CAELO_EXTRACT_NO_DEDUP);
}

int dedup_launch(uint64_t *bits, void *scratch, bool enabled, hipStream_t s) {
    caelo_frame_set fs = {};
    fs.n = 1;
    fs.f[0].bits = (unsigned long long *)bits;
    fs.f[0].dd = (DedupScratch *)scratch;
    return dedup_set(fs, enabled, s);
}

int dedup_set(const caelo_frame_set &fs, bool enabled, hipStream_t s) {
    if (enabled) {  // the caller asked dedup_enabled() and let k_patches fill the tables
        k_dd_verify<<<dim3(CAELO_FRAME_PATCHES / 4, 1, fs.n), 256, 0, s>>>(fs);
        CAELO_LAUNCH_CHECK();
    }
    k_dd_scan<<<dim3(1, 1, fs.n), 1024, 0, s>>>(fs, enabled ? 0 : 1);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}
