// match.hip -- brute-force descriptor matching, rigid fit, RANSAC pose scoring.
//
// Reference behaviour restated here (never its code):
//   NN match             Match.py:257-258   cdist (f64) + argmin(axis=0), first minimum wins
//   SolveRT              Match.py:138-158   Kabsch via 3x3 SVD, reflection quirk at :151-155
//   RANSAC4RT            Match.py:162-218   4-point samples with replacement, 100..500 trials,
//                                            threshold escalation 0.4 -> 0.8 -> 1.6
//   SolveRelativePose    Match.py:260-283   inlier refit
//
// The reference draws from NumPy's global RNG inside the loop; here the caller hands over the
// uniform doubles in consumption order (3 levels x 500 trials x 4), every hypothesis of a level is
// scored in parallel (one wavefront each, pairs staged in LDS, ballot + popcount for the inlier count) and
// the last workgroup of the launch replays the sequential accept / early-exit rules over the count array
// as a prefix maximum (verified equivalent on the reference: SURVEY 8a-9), writes the inlier mask and refits.
// Also here: k_icp_nn / k_icp_fit_apply, one iteration of the reference's point-to-point ICP.
// The workspaces of caelo_match / caelo_ransac are self-cleaning (zero-filled once by their owner).
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "caelo_internal.h"
#include "caelo_rigid.h"

#define MT_MAXDIM 64

// ------------------------------------------------------------------------------------------------
// NN match, fast path: the all-pairs matrix on the f64 matrix cores, the argmin certified afterwards.
//   v[i][j] = |f0_i|^2 - 2 <f0_i, f1_j>  (= d^2 - |f1_j|^2) from v_mfma_f64_16x16x4_f64, with the
//   rigorous rounding bound e[i][j] = kappa (|f0_i|^2 + |f1_j|^2), kappa = (4 dim + 64) 2^-53.
//   The exact argmin i* of SciPy's cdist satisfies v[i*] - e[i*] <= min_i (v[i] + e[i]), so only rows
//   passing that test can win; they are re-evaluated exactly like cdist (sequential f64 sum of squared
//   differences, sqrt) and the first minimum is kept (Match.py:257-258).  With f64 products the window
//   is ~1e-13 wide: one row per column survives unless descriptors are duplicated.
//   (An f32 MFMA version of the same filter keeps ~100 rows per column on these descriptors -- the
//   |a|^2+|b|^2-2ab form cancels ~4 digits -- and was slower than the plain f64 scan.)
// Grid = (blocks of 2 column tiles = 32 frame-1 descriptors) x pairs of the set: 256 workgroups for 8 pairs, one per CU.
// A workgroup is 8 wavefronts: wavefront w owns column tile w & 1 (its B fragments stay in registers) and the 16-row
// tiles t of frame 0 with t % 4 == w >> 1.  The workgroup walks frame 0 ONCE: each step stages 8 row tiles (16 x 64 f32
// each) in LDS with four coalesced 16-byte loads per thread, in flight during the previous step's MFMAs, double buffered,
// and every wavefront reads its A fragments from there -- frame 0 is read from L2 once per 32 columns (8.4 MB per pair)
// instead of once per 16 (16.7 MB).
// The f64 MFMA is the slow one on this chip (16x16x4: 64 cycles of the pipe, SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA;
// peak 78.6 TFLOP/s) and a tile needs 16 of them and ~200 VALU instructions (f32 -> f64, norms, the top-3 bookkeeping):
// the two wavefronts of a SIMD run the loop in opposite phase (one in its MFMAs while the other does the bookkeeping of
// its previous tile).  Measured: 49 us per 8 pairs; 30 us with the MFMAs replaced by plain FMAs, 14.5 us is the MFMA time
// alone -- the f64 matrix instruction and the f64 VALU share their ALUs on this chip (matrix f64 peak = vector f64 peak
// = 78.6 TFLOP/s), so f64 bookkeeping next to f64 MFMAs adds up instead of overlapping; the remaining lever is fewer
// f64 VALU instructions per tile.
// Partial top-3 lists meet in LDS, one thread per column certifies.  Nothing crosses workgroups (round 1 split the rows
// over four 1024-thread workgroups per column tile that met through global tickets and starved behind the encoder's
// persistent grids).
// ------------------------------------------------------------------------------------------------
typedef double mm_f64x4 __attribute__((ext_vector_type(4)));
#define MM_WAVES 8
#define MM_CT 2        // column tiles per workgroup
#define MM_RQ 4        // row quarters: wavefront w owns column tile w & 1 and the row tiles t with t % 4 == w >> 1
#define MM_KSTEPS 16   // dim <= 64
#define MM_LDA 68      // floats per staged row (64 + 4: rows start 4 banks apart)
#define MM_TPS 2       // row tiles per wavefront and step: a step stages MM_RQ * MM_TPS tiles (128 rows), one barrier each
#define MM_STEP_TILES (MM_RQ * MM_TPS)

static inline int64_t ms_ws_bytes(int64_t kmax);
CAELO_API int64_t caelo_match_ws_bytes(int64_t k_max) {
    // [0] columns re-scanned exactly, [1] columns decided between two rows (statistics only), then the f16 operand images of
    // both frames (match_screen.inc); k_max = the larger of the two frames' row capacities
    return ms_ws_bytes(k_max > 0 ? k_max : 1);
}

__device__ inline double exact_dist(const float *a, const float *b, int dim) {
    double acc = 0.0;
    for (int c = 0; c < dim; ++c) {
        const double d = __dsub_rn((double)a[c], (double)b[c]);
        acc = __dadd_rn(acc, __dmul_rn(d, d));
    }
    return sqrt(acc);
}

// ---- top-3 with the row index INSIDE the key.  The lower bound lo of a row is widened by 2^-30 (|v| + e) and its low
// MM_IDX_BITS mantissa bits are replaced by the row index (a change of < 2^-31 |lo|: the key is still a lower bound).  An
// ascending top-3 of such keys is a five-instruction min / max network -- no compares, no selects, no separate index registers
// (the compiler turned the select form into branches and register shuffles: 225 VALU instructions per tile, now ~100).
#define MM_IDX_BITS 21
#define MM_IDX_MASK ((1 << MM_IDX_BITS) - 1)
__device__ inline double mm_key(double lo, int i) {
    return __longlong_as_double((__double_as_longlong(lo) & ~(long long)MM_IDX_MASK) | (long long)i);
}
__device__ inline int mm_key_index(double key) { return (int)(__double_as_longlong(key) & (long long)MM_IDX_MASK); }
__device__ inline void mm_key_insert(double key, double &L1, double &L2, double &L3) {
    const double u1 = __builtin_fmax(L1, key);
    L1 = __builtin_fmin(L1, key);
    const double u2 = __builtin_fmax(L2, u1);
    L2 = __builtin_fmin(L2, u1);
    L3 = __builtin_fmin(L3, u2);
}

// 16 channels [16g, 16g+16) of one descriptor row as doubles (zero beyond dim / for an invalid row)
template <bool VEC>
__device__ inline void load_frag(const float *row, bool valid, int g, int dim, double out[MM_KSTEPS]) {
    if (VEC) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 16 * g + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid && c < dim) v = *(const float4 *)(row + c);  // dim % 4 == 0 on this path
            out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int s = 0; s < MM_KSTEPS; ++s) {
            const int c = 16 * g + s;
            out[s] = (valid && c < dim) ? (double)row[c] : 0.0;
        }
    }
}

// this thread's 16 bytes of a staged row tile: row (tid >> 4) & 15 of tile `t`, channels 4 (tid & 15) .. + 3
template <bool VEC>
__device__ inline float4 mm_stage_load(const float *f0, int ld0, int k0, int dim, int t, int tid) {
    const int row = (t << 4) + ((tid >> 4) & 15), c = (tid & 15) * 4;
    // rows past k0 (the tail of the last tile, whole tiles of the last step) are staged as a point 1e18 away on channel 0: their
    // distance to anything is ~1e36, so they never win a column and the bookkeeping needs no validity test
    float4 v = make_float4(c == 0 ? 1.0e18f : 0.f, 0.f, 0.f, 0.f);
    if (row < k0) {
        v.x = 0.f;
        const float *src = f0 + (size_t)row * ld0 + c;
        if (VEC) { if (c < dim) v = *(const float4 *)src; }  // dim % 4 == 0 on this path
        else {
            if (c < dim) v.x = src[0];
            if (c + 1 < dim) v.y = src[1];
            if (c + 2 < dim) v.z = src[2];
            if (c + 3 < dim) v.w = src[3];
        }
    }
    return v;
}

template <bool VEC>
__global__ void __launch_bounds__(64 * MM_WAVES) k_match_mfma(const caelo_pair_set ps, int ld0, int64_t k0_max, int ld1,
                                                              int64_t k1_max, int dim) {
    const caelo_pair_dev &P = ps.p[blockIdx.z];
    const float *__restrict__ f0 = P.f0, *__restrict__ f1 = P.f1;
    const int32_t *n0p = P.n0, *n1p = P.n1;
    int64_t *__restrict__ pair_idx = P.pair_idx;
    int32_t *stats = (int32_t *)P.ws_match;
    __shared__ __attribute__((aligned(16))) float sA[2][MM_STEP_TILES][16 * MM_LDA];  // [buffer][row tile of the step][row][channel]
    __shared__ double sL[3][MM_WAVES][16];
    __shared__ double sU[MM_WAVES][16];
    __shared__ int s_rescan[16 * MM_CT];
    __shared__ double s_rd[MM_WAVES];
    __shared__ int s_ri[MM_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, x = lane & 15;
    const int cw = wave & (MM_CT - 1), rh = wave / MM_CT;
    // counts live on the device; clamp so that a caller who forgot to order this launch after the
    // producer of n0/n1 reads garbage rows, never out of bounds
    const int k0 = n0p ? min(max(*n0p, 0), (int)k0_max) : (int)k0_max;
    const int k1 = n1p ? min(max(*n1p, 0), (int)k1_max) : (int)k1_max;
    const int jb = blockIdx.x * 16 * MM_CT;  // first column of the workgroup
    if (jb >= k1) return;
    if (k0 == 0) {         // no frame-0 descriptor at all (the reference's argmin would raise): index 0, the pose fails
        if (tid < 16 * MM_CT && jb + tid < k1) pair_idx[jb + tid] = 0;
        return;
    }
    const int j0 = jb + cw * 16;
    // (a wavefront whose column tile lies beyond k1 computes on zeros: same instruction stream, its results are never read)
    const double kappa = (4.0 * (double)dim + 64.0) * 1.1102230246251565e-16;  // >= 2x the worst-case bound (dim + 20) 2^-53
    const double BIG = 1.0e300;
    // B fragments (this column tile) and |f1_j|^2.  k-step s of lane group g <-> channel 16 g + s.
    double b[MM_KSTEPS];
    load_frag<VEC>(f1 + (size_t)(j0 + x) * ld1, j0 + x < k1, g, dim, b);
    double n1 = 0.0;
#pragma unroll
    for (int s = 0; s < MM_KSTEPS; ++s) n1 += b[s] * b[s];
    n1 += __shfl_xor(n1, 16);
    n1 += __shfl_xor(n1, 32);
    double L1 = BIG, L2 = BIG, L3 = BIG, U = BIG;  // keys (mm_key): BIG carries no index
    const int ntiles = (k0 + 15) >> 4;
    const int nsteps = (ntiles + MM_STEP_TILES - 1) / MM_STEP_TILES;
    // thread -> (tiles st_par, st_par + 2, ... of a step, position inside the tile): 4 loads of 16 B in flight per thread
    const int st_par = tid >> 8;
    float *st_dst0 = &sA[0][st_par][((tid >> 4) & 15) * MM_LDA + (tid & 15) * 4];
    float4 stage[MM_STEP_TILES / 2];
#pragma unroll
    for (int k = 0; k < MM_STEP_TILES / 2; ++k) {
        stage[k] = mm_stage_load<VEC>(f0, ld0, k0, dim, 2 * k + st_par, tid);
        *(float4 *)(st_dst0 + 2 * k * 16 * MM_LDA) = stage[k];
    }
    __syncthreads();
    // bookkeeping of one finished tile: rows i0 + g + 4 r of the accumulator against this lane's column
#define MM_BOOKKEEP(ACC, ACC2, PN, I0)                                                              \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                 \
        const int row = g + 4 * r; /* f64 C/D layout: row = (lane >> 4) + 4 * reg, col = lane & 15 */ \
        const double n0 = __shfl(PN, row); /* lane `row` (g = 0) holds that row's norm */           \
        const double v = __builtin_fma(-2.0, (ACC)[r] + (ACC2)[r], n0); /* = n0 - 2 dot, one rounding */ \
        const double e = kappa * (n0 + n1);                                                         \
        U = __builtin_fmin(U, v + e);                                                               \
        const double ew = __builtin_fma(__builtin_fabs(v) + e, 9.313225746154785e-10 /* 2^-30 */, e); \
        mm_key_insert(mm_key(v - ew, (I0) + row), L1, L2, L3);                                      \
    }
    mm_f64x4 accP = {0.0, 0.0, 0.0, 0.0}, acc2P = {0.0, 0.0, 0.0, 0.0};
    double pP = 0.0;
    int i0P = -1;  // no tile yet
    // The two wavefronts of a SIMD (w and w + 4) run the loop in opposite phase: one issues a tile's MFMAs and then the
    // bookkeeping of the tile before, the other the bookkeeping first -- while one occupies the matrix pipe the other
    // has VALU work, and the barrier at the end of a step re-aligns them to exactly that.
    auto main_loop = [&](auto vfirst_t) {
        constexpr bool VFIRST = decltype(vfirst_t)::value;
#pragma unroll 1
        for (int it = 0; it < nsteps; ++it) {
            const int buf = it & 1;
            if (it + 1 < nsteps) {  // the next step's rows: in flight during this step's MFMAs
#pragma unroll
                for (int k = 0; k < MM_STEP_TILES / 2; ++k)
                    stage[k] = mm_stage_load<VEC>(f0, ld0, k0, dim, MM_STEP_TILES * (it + 1) + 2 * k + st_par, tid);
            }
#pragma unroll
            for (int k = 0; k < MM_TPS; ++k) {
                if (VFIRST) {
                    if (i0P >= 0) { MM_BOOKKEEP(accP, acc2P, pP, i0P) }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const int t = MM_STEP_TILES * it + k * MM_RQ + rh;  // tiles past the last hold zeros and fail i < k0
                const float *ar = &sA[buf][k * MM_RQ + rh][x * MM_LDA + 16 * g];
                double a[MM_KSTEPS];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *(const float4 *)(ar + 4 * q);
                    a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
                }
                double p = 0.0;
#pragma unroll
                for (int s = 0; s < MM_KSTEPS; ++s) p = __builtin_fma(a[s], a[s], p);
                p += __shfl_xor(p, 16);
                p += __shfl_xor(p, 32);  // |f0_{i0+x}|^2 on every lane with this x
                // two accumulators: no MFMA waits for its predecessor (the rounding bound kappa holds for any summation order)
                mm_f64x4 acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < MM_KSTEPS; s += 2) {
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s + 1], b[s + 1], acc2, 0, 0, 0);
                }
                if (!VFIRST) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (i0P >= 0) { MM_BOOKKEEP(accP, acc2P, pP, i0P) }  // the previous tile: its MFMAs finished long ago
                }
                accP = acc; acc2P = acc2; pP = p; i0P = t << 4;
            }
            if (it + 1 < nsteps) {  // the other buffer: nobody reads it in this step
#pragma unroll
                for (int k = 0; k < MM_STEP_TILES / 2; ++k)
                    *(float4 *)(st_dst0 + ((buf ^ 1) * MM_STEP_TILES + 2 * k) * 16 * MM_LDA) = stage[k];
            }
            __syncthreads();
        }
    };
    if (__builtin_amdgcn_readfirstlane(wave) < 4) main_loop(std::false_type{});
    else main_loop(std::true_type{});
    if (i0P >= 0) { MM_BOOKKEEP(accP, acc2P, pP, i0P) }
#undef MM_BOOKKEEP
    // ---- workgroup top-3 per column: merge the 4 lane groups by shuffles, the four row quarters through LDS
#define MM_SHFL_MERGE(OFF)                                                                           \
    {                                                                                                \
        const double pL1 = __shfl_xor(L1, OFF), pL2 = __shfl_xor(L2, OFF), pL3 = __shfl_xor(L3, OFF); \
        U = __builtin_fmin(U, __shfl_xor(U, OFF));                                                   \
        mm_key_insert(pL1, L1, L2, L3);                                                              \
        mm_key_insert(pL2, L1, L2, L3);                                                              \
        mm_key_insert(pL3, L1, L2, L3);                                                              \
    }
    MM_SHFL_MERGE(16)
    MM_SHFL_MERGE(32)
    if (g == 0) {
        sL[0][wave][x] = L1; sL[1][wave][x] = L2; sL[2][wave][x] = L3;
        sU[wave][x] = U;
    }
    __syncthreads();
    // ---- merge the row quarters and certify: one thread per column of the workgroup's 32
    if (tid < 16 * MM_CT) {
        const int ct = tid >> 4, col = tid & 15;
        const int j = jb + tid;
        int rescan = 0;
        if (j < k1) {
            double a1 = BIG, a2 = BIG, a3 = BIG, Umin = BIG;
#pragma unroll
            for (int h = 0; h < MM_WAVES / MM_CT; ++h) {
                const int w = ct + MM_CT * h;
                Umin = __builtin_fmin(Umin, sU[w][col]);
                mm_key_insert(sL[0][w][col], a1, a2, a3);
                mm_key_insert(sL[1][w][col], a1, a2, a3);
                mm_key_insert(sL[2][w][col], a1, a2, a3);
            }
            const int i1 = mm_key_index(a1), i2 = mm_key_index(a2);
            if (a3 <= Umin) {
                rescan = 1;  // three or more rows inside the window
                if (stats) atomicAdd(&stats[0], 1);
            } else if (a2 <= Umin) {
                if (stats) atomicAdd(&stats[1], 1);
                const float *bj = f1 + (size_t)j * ld1;
                const double d1 = exact_dist(f0 + (size_t)i1 * ld0, bj, dim), d2 = exact_dist(f0 + (size_t)i2 * ld0, bj, dim);
                pair_idx[j] = (d2 < d1 || (d2 == d1 && i2 < i1)) ? i2 : i1;
            } else {
                pair_idx[j] = i1;  // certified without an exact evaluation
            }
        }
        s_rescan[tid] = rescan;
    }
    __syncthreads();
    // ---- exact re-scan of a column whose window holds three or more rows (whole workgroup)
    for (int cidx = 0; cidx < 16 * MM_CT; ++cidx) {
        if (!s_rescan[cidx]) continue;  // uniform
        const float *bj = f1 + (size_t)(jb + cidx) * ld1;
        double best = BIG;
        int besti = 0x7FFFFFFF;
        for (int i = tid; i < k0; i += 64 * MM_WAVES) {
            const double dd = exact_dist(f0 + (size_t)i * ld0, bj, dim);
            if (dd < best) { best = dd; besti = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(best, o);
            const int obi = __shfl_xor(besti, o);
            if (ob < best || (ob == best && obi < besti)) { best = ob; besti = obi; }
        }
        __syncthreads();
        if (lane == 0) { s_rd[wave] = best; s_ri[wave] = besti; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < MM_WAVES; ++w)
                if (s_rd[w] < best || (s_rd[w] == best && s_ri[w] < besti)) { best = s_rd[w]; besti = s_ri[w]; }
            pair_idx[jb + cidx] = besti;
        }
    }
}

#include "match_screen.inc"

CAELO_API int caelo_match(caelo_ctx *c, const float *f0, int ld0, int64_t k0_max, const int32_t *n0, const float *f1,
                          int ld1, int64_t k1_max, const int32_t *n1, int dim, int64_t *pair_idx, void *ws, void *stream) {
    CAELO_REQUIRE(c && f0 && f1 && pair_idx && ws, "null argument");
    caelo_pair_set ps = {};
    ps.n = 1;
    ps.p[0].f0 = f0; ps.p[0].n0 = n0; ps.p[0].f1 = f1; ps.p[0].n1 = n1; ps.p[0].pair_idx = pair_idx; ps.p[0].ws_match = ws;
    return match_set(ps, ld0, k0_max, ld1, k1_max, dim, caelo_stream(stream));
}

// The pipeline's launch shape of the NN match -- `n_pairs` pairs of consecutive frame rows ([1024][64] f32: descriptor 0:60) behind
// ONE k_match_prep + ONE k_match_screen launch -- repeated between HIP events on `stream`: ms_host[0] = both kernels, [1] =
// k_match_prep alone, averaged over `repeats` (bench.py's second roofline object; tools/roofline_launch.py for rocprofv3).
CAELO_API int caelo_match_profile(caelo_ctx *c, const float *const *rows, int n_pairs, const int32_t *const *n_key, int64_t *pair_idx,
                                  void *ws, int repeats, void *stream, float *ms_host) {
    CAELO_REQUIRE(c && rows && pair_idx && ws && ms_host && n_pairs >= 1 && n_pairs <= CAELO_FB_MAX && repeats >= 1, "bad argument");
    hipStream_t s = caelo_stream(stream);
    caelo_pair_set ps = {};
    ps.n = n_pairs;
    const size_t wsb = (size_t)caelo_match_ws_bytes(CAELO_MAX_KEYPTS);
    for (int i = 0; i < n_pairs; ++i) {
        ps.p[i].f0 = rows[i]; ps.p[i].f1 = rows[i + 1];
        ps.p[i].n0 = n_key ? n_key[i] : nullptr; ps.p[i].n1 = n_key ? n_key[i + 1] : nullptr;
        ps.p[i].pair_idx = pair_idx + (size_t)i * CAELO_MAX_KEYPTS;
        ps.p[i].ws_match = (char *)ws + (size_t)i * wsb;
    }
    hipEvent_t ev[3];
    for (hipEvent_t &e : ev) CAELO_HIP(hipEventCreate(&e));
    const int64_t kpad = ms_pad16(CAELO_MAX_KEYPTS);
    int rc = CAELO_OK;
    float both = 0.f, prep = 0.f;
    for (int r = 0; r < repeats && rc == CAELO_OK; ++r) {
        CAELO_HIP(hipEventRecord(ev[0], s));
        k_match_prep<<<dim3((unsigned)((kpad / 16 + 3) / 4), 1, ps.n), 256, 0, s>>>(ps, 64, CAELO_MAX_KEYPTS, 60, kpad, 1);
        CAELO_HIP(hipEventRecord(ev[1], s));
        k_match_screen<<<dim3((unsigned)((CAELO_MAX_KEYPTS + 16 * MS_CT - 1) / (16 * MS_CT)), 1, ps.n), 64 * MS_NW, 0, s>>>(ps, 64, CAELO_MAX_KEYPTS, 64,
                                                                                                                 CAELO_MAX_KEYPTS, 60, kpad, 1);
        CAELO_HIP(hipEventRecord(ev[2], s));
        CAELO_HIP(hipEventSynchronize(ev[2]));
        float a = 0.f, b = 0.f;
        CAELO_HIP(hipEventElapsedTime(&a, ev[0], ev[2]));
        CAELO_HIP(hipEventElapsedTime(&b, ev[0], ev[1]));
        both += a; prep += b;
    }
    for (hipEvent_t &e : ev) (void)hipEventDestroy(e);
    ms_host[0] = both / (float)repeats;
    ms_host[1] = prep / (float)repeats;
    return rc;
}

int match_set(const caelo_pair_set &ps, int ld0, int64_t k0_max, int ld1, int64_t k1_max, int dim, hipStream_t s) {
    CAELO_REQUIRE(ps.n >= 1 && ps.n <= CAELO_FB_MAX, "bad pair count");
    CAELO_REQUIRE(dim > 0 && dim <= MT_MAXDIM && ld0 >= dim && ld1 >= dim && k0_max > 0 && k1_max > 0, "bad shape");
    CAELO_REQUIRE(k0_max + 16 * MM_STEP_TILES < MM_IDX_MASK, "too many frame-0 rows (the row index travels in 21 key bits)");
    // the f16 screen + exact certification (match_screen.inc) for every descriptor width that leaves room for the two norm slots in
    // K = 64 (the reference's descriptors are 60 wide); wider ones: the all-f64 kernel.  Same pair_idx bit for bit
    // (tests/test_gpu_parity.py::test_match_shape_sweep_vs_oracle covers both).
    if (dim <= 62) {
        const int64_t kpad = ms_pad16(k0_max > k1_max ? k0_max : k1_max);
        bool v4 = (dim % 4 == 0) && (ld0 % 4 == 0) && (ld1 % 4 == 0);
        for (int i = 0; i < ps.n; ++i) v4 = v4 && (((uintptr_t)ps.p[i].f0 | (uintptr_t)ps.p[i].f1) & 15u) == 0;
        k_match_prep<<<dim3((unsigned)((kpad / 16 + 3) / 4), 1, ps.n), 256, 0, s>>>(ps, ld0, k0_max, dim, kpad, v4 ? 1 : 0);
        CAELO_LAUNCH_CHECK();
        k_match_screen<<<dim3((unsigned)((k1_max + 16 * MS_CT - 1) / (16 * MS_CT)), 1, ps.n), 64 * MS_NW, 0, s>>>(ps, ld0, k0_max, ld1, k1_max,
                                                                                                            dim, kpad, v4 ? 1 : 0);
        CAELO_LAUNCH_CHECK();
        return CAELO_OK;
    }
    const int64_t tiles = (k1_max + 16 * MM_CT - 1) / (16 * MM_CT);  // workgroups of MM_CT column tiles
    bool vec = (dim % 4 == 0) && (ld0 % 4 == 0) && (ld1 % 4 == 0);
    for (int i = 0; i < ps.n; ++i) vec = vec && (((uintptr_t)ps.p[i].f0 | (uintptr_t)ps.p[i].f1) & 15u) == 0;
    dim3 grid((unsigned)tiles, 1, ps.n);
    if (vec)
        k_match_mfma<true><<<grid, 64 * MM_WAVES, 0, s>>>(ps, ld0, k0_max, ld1, k1_max, dim);
    else
        k_match_mfma<false><<<grid, 64 * MM_WAVES, 0, s>>>(ps, ld0, k0_max, ld1, k1_max, dim);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

// residual of Match.py:191-192 in f32
__device__ inline float residual(const float *R, const float *T, float ax, float ay, float az, float bx, float by, float bz) {
    const float px = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(R[0], bx), __fmul_rn(R[1], by)), __fmul_rn(R[2], bz)), T[0]);
    const float py = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(R[3], bx), __fmul_rn(R[4], by)), __fmul_rn(R[5], bz)), T[1]);
    const float pz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(R[6], bx), __fmul_rn(R[7], by)), __fmul_rn(R[8], bz)), T[2]);
    const float dx = __fsub_rn(ax, px), dy = __fsub_rn(ay, py), dz = __fsub_rn(az, pz);
    return sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
}

// ------------------------------------------------------------------------------------------------
// rigid fit over n pairs (one workgroup): single pass accumulating count, sums and the raw
// cross-moments in f64, H = sum p1 p0^T - n m1 m0^T, then the 3x3 SVD on one thread.
// pair i = (p0[idx0 ? idx0[i] : i], p1[i]) restricted to mask.
// ------------------------------------------------------------------------------------------------
#define FIT_TERMS 16

__device__ void fit_block(const float *p0, int ld0, const int64_t *idx0, const float *p1, int ld1, const uint8_t *mask,
                          int n, float *R, float *T, int *credible) {
    __shared__ double red[4][FIT_TERMS];
    const int tid = threadIdx.x;
    double a[FIT_TERMS];
#pragma unroll
    for (int t = 0; t < FIT_TERMS; ++t) a[t] = 0.0;
    for (int i = tid; i < n; i += blockDim.x) {
        if (mask && !mask[i]) continue;
        const float *u = p0 + (size_t)ld0 * (idx0 ? idx0[i] : i);
        const float *v = p1 + (size_t)ld1 * i;
        const double x0 = u[0], y0 = u[1], z0 = u[2], x1 = v[0], y1 = v[1], z1 = v[2];
        a[0] += 1.0;
        a[1] += x0; a[2] += y0; a[3] += z0;
        a[4] += x1; a[5] += y1; a[6] += z1;
        a[7] += x1 * x0; a[8] += x1 * y0; a[9] += x1 * z0;   // P1^T P0  (:146)
        a[10] += y1 * x0; a[11] += y1 * y0; a[12] += y1 * z0;
        a[13] += z1 * x0; a[14] += z1 * y0; a[15] += z1 * z0;
    }
#pragma unroll
    for (int t = 0; t < FIT_TERMS; ++t)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a[t] += __shfl_xor(a[t], o);
    __syncthreads();
    if ((tid & 63) == 0)
#pragma unroll
        for (int t = 0; t < FIT_TERMS; ++t) red[tid >> 6][t] = a[t];
    __syncthreads();
    if (tid == 0) {
        double s[FIT_TERMS];
        for (int t = 0; t < FIT_TERMS; ++t) {
            s[t] = 0.0;
            for (unsigned w = 0; w < blockDim.x / 64; ++w) s[t] += red[w][t];
        }
        if (s[0] >= 1.0) {
            const double cnt = s[0];
            const double m0[3] = {s[1] / cnt, s[2] / cnt, s[3] / cnt}, m1[3] = {s[4] / cnt, s[5] / cnt, s[6] / cnt};
            double H[9];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) H[3 * i + j] = s[7 + 3 * i + j] - cnt * m1[i] * m0[j];
            const int cred = rigid_from_H(H, m0, m1, R, T);
            if (credible) *credible = cred;
        }
    }
}

__global__ void __launch_bounds__(256) k_solve_rt(const float *p0, const float *p1, int n, float *R, float *T, int *credible) {
    fit_block(p0, 3, nullptr, p1, 3, nullptr, n, R, T, credible);
}

CAELO_API int caelo_solve_rt(caelo_ctx *c, const float *p0, const float *p1, int64_t n, float *R, float *T,
                             int32_t *credible, void *stream) {
    CAELO_REQUIRE(c && p0 && p1 && R && T && n > 0, "bad argument");
    k_solve_rt<<<1, 256, 0, caelo_stream(stream)>>>(p0, p1, (int)n, R, T, credible);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

// ------------------------------------------------------------------------------------------------
// RANSAC
// ------------------------------------------------------------------------------------------------
// workspace of one pair: the inlier counts of the 500 hypotheses of the level being replayed.  Written by one kernel,
// read by the next (k_ransac_hyp -> k_ransac_finish) or inside one workgroup: nothing to initialise, nothing to clean.
struct RansacWs {
    int32_t counts[CAELO_RANSAC_MAX_TRIALS];
};

CAELO_API int64_t caelo_ransac_ws_bytes(void) { return (int64_t)sizeof(RansacWs); }

struct RansacVerdict {
    int iterations, success, best_trial;  // best_trial within the level, -1 = none accepted
};

// sequential accept / exit rules of Match.py:166-169,:181,:195-214 replayed over the counts (one wave:
// the counts are staged in LDS, lane 0 walks them)
__device__ __forceinline__ void ransac_replay(int N, const int32_t *counts, RansacVerdict *out, const int *preloaded = nullptr) {
    // The loop of :181-206 keeps the running maximum of the admissible counts (strict >: the FIRST
    // occurrence wins) and stops at the first iteration it >= 100 whose running maximum reached
    // 0.25 N, or at 500.  Restated as a prefix-max scan so one wavefront does it in parallel:
    //   M[i]  = max(c'[0..i]),  c'[i] = counts[i] if counts[i] >= leastInliers else 0
    //   it*   = 1 + min{ i >= 99 : M[i] >= 0.25 N }   (500 if none)
    //   best  = min{ j < it* : c'[j] == M[it*-1] }    (none if M[it*-1] == 0 -> the level failed)
    const int lane = threadIdx.x;
    const int least = (100 < (int)(0.2 * N)) ? 100 : (int)(0.2 * N);  // :166
    const double min_success = 0.25 * N;                              // :167
    const int PER = (CAELO_RANSAC_MAX_TRIALS + 63) / 64;              // 8 consecutive trials per lane
    int c[(CAELO_RANSAC_MAX_TRIALS + 63) / 64], pm[(CAELO_RANSAC_MAX_TRIALS + 63) / 64];
    int run = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int i = lane * PER + q;
        int v = preloaded ? preloaded[q] : (i < CAELO_RANSAC_MAX_TRIALS ? counts[i] : 0);
        v = v >= least ? v : 0;
        c[q] = v;
        run = run > v ? run : v;
        pm[q] = run;
    }
    // exclusive prefix max of the lane maxima
    int incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl = incl > up ? incl : up;
    }
    int excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 0;
    int stop = CAELO_RANSAC_MAX_TRIALS;  // it*
#pragma unroll
    for (int q = PER - 1; q >= 0; --q) {
        const int i = lane * PER + q;
        pm[q] = pm[q] > excl ? pm[q] : excl;
        if (i >= 99 && i < CAELO_RANSAC_MAX_TRIALS && (double)pm[q] >= min_success) stop = i + 1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(stop, o); stop = stop < t ? stop : t; }
    // running maximum after `stop` iterations = M[stop-1]
    int target = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q)
        if (lane * PER + q == stop - 1) target = pm[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(target, o); target = target > t ? target : t; }
    int best = 0x7FFFFFFF;
#pragma unroll
    for (int q = PER - 1; q >= 0; --q) {
        const int i = lane * PER + q;
        if (i < stop && target > 0 && c[q] == target) best = i;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(best, o); best = best < t ? best : t; }
    if (lane != 0) return;
    // leastInliers == 0 (N < 5) admits every hypothesis.  N == 0 -- a frame without a single key point, where the reference's
    // sampling raises IndexError (Match.py:182-187) -- is reported as a failed pose, not as a success over nothing.
    const int success = N > 0 && ((target > 0) || (least <= 0));
    out->iterations = stop;
    out->success = success;
    out->best_trial = (success && target > 0) ? best : -1;  // N < 5: success without any accepted hypothesis -> identity (:177)
}

// the four sampled pairs of a hypothesis as SolveRT sees them (Match.py:141-146; means / centring in f32 like np.mean on
// f32 rows, covariance in f64)
struct Sample4 {
    double m0[3], m1[3], H[9];
    float c0[4][3], c1[4][3];
    int idx[4];
};
__device__ inline void sample4(const float *P0, int l0, const int64_t *pidx, const float *P1, int l1, int N, const double *r4, Sample4 &s) {
    float s0[4][3], s1[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = (int)(r4[q] * (double)N);  // :182-184 idx = int32(u * N), with replacement
        s.idx[q] = idx;
        const float *a = P0 + (size_t)l0 * (pidx ? pidx[idx] : idx);
        const float *b = P1 + (size_t)l1 * idx;
#pragma unroll
        for (int c = 0; c < 3; ++c) { s0[q][c] = a[c]; s1[q][c] = b[c]; }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float mm0 = __fdiv_rn(__fadd_rn(__fadd_rn(__fadd_rn(s0[0][a], s0[1][a]), s0[2][a]), s0[3][a]), 4.0f);
        const float mm1 = __fdiv_rn(__fadd_rn(__fadd_rn(__fadd_rn(s1[0][a], s1[1][a]), s1[2][a]), s1[3][a]), 4.0f);
        s.m0[a] = mm0; s.m1[a] = mm1;
#pragma unroll
        for (int q = 0; q < 4; ++q) { s.c0[q][a] = __fsub_rn(s0[q][a], mm0); s.c1[q][a] = __fsub_rn(s1[q][a], mm1); }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double h = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) h += (double)s.c1[q][i] * (double)s.c0[q][j];
            s.H[3 * i + j] = h;
        }
}

// hypothesis from 4 sampled pairs (SolveRT on the sample, Match.py:141-157)
__device__ inline void sample_hypothesis(const float *P0, int l0, const int64_t *pidx, const float *P1, int l1, int N,
                                         const double *r4, float R[9], float T[3]) {
    float s0[4][3], s1[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = (int)(r4[q] * (double)N);  // :182-184 idx = int32(u * N), with replacement
        const float *a = P0 + (size_t)l0 * (pidx ? pidx[idx] : idx);
        const float *b = P1 + (size_t)l1 * idx;
#pragma unroll
        for (int c = 0; c < 3; ++c) { s0[q][c] = a[c]; s1[q][c] = b[c]; }
    }
    double m0[3], m1[3], H[9];
    float c0[4][3], c1[4][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float mm0 = __fdiv_rn(__fadd_rn(__fadd_rn(__fadd_rn(s0[0][a], s0[1][a]), s0[2][a]), s0[3][a]), 4.0f);
        const float mm1 = __fdiv_rn(__fadd_rn(__fadd_rn(__fadd_rn(s1[0][a], s1[1][a]), s1[2][a]), s1[3][a]), 4.0f);
        m0[a] = mm0; m1[a] = mm1;
#pragma unroll
        for (int q = 0; q < 4; ++q) { c0[q][a] = __fsub_rn(s0[q][a], mm0); c1[q][a] = __fsub_rn(s1[q][a], mm1); }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double h = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) h += (double)c1[q][i] * (double)c0[q][j];
            H[3 * i + j] = h;
        }
    rigid_from_H<false>(H, m0, m1, R, T);   // (hypotheses: see rigid_from_H_jacobi)
}

// ---- how far the reference's own arithmetic can move a residual of this hypothesis (the certificate of caelo.h) ----------
// The reference forms H = P1^T P0 in float32 through its BLAS (rounding error <= ~4 eps32 sum |c1||c0| per entry, order and
// fusing unknown), takes the float64 SVD of THAT matrix, rounds U and Vh to float32, multiplies R and T in float32 and
// evaluates the residuals in float32 (Match.py:146-157,:190-192); this kernel fits the exact float64 covariance of the same
// centred float32 points.  The rotation of a Kabsch fit moves by at most 2 |dH| / (s2 + s3) under a perturbation dH of the
// covariance, so with kappa = s1 / s2 the two poses differ by O(eps32 (1 + kappa)) in R, and the residual of pair j by at most
//     band_j = CB eps32 [ (1 + kappa) |p1_j - mean1| + |p0_j| + |p1_j| + |T| ],      CB = 16
// (measured over 54 000 samples of the golden pairs against NumPy: the largest ratio residual difference / bracket is 3.5,
// the 99.9th percentile 2.7; CB = 16 leaves a factor 4.5.  The constant is CALIBRATED, not proven, so the bound is checked wherever
// a true count is known: tools/parity_soak.py compares every hypothesis count of the oracle's first-level loops with `hi`
// (profiles/r0x_parity_soak*.txt: 0 violations in > 400 k counts), and the host half compares every count it evaluates with its
// bound and, on a violation, decides the pair without bounds and counts the event -- caelo_host_bound_violations, certify.hip).
// kappa is bounded from the
// invariants I1 = |H|_F^2, I2 = |cof H|_F^2: kappa <= sqrt(2) I1 / sqrt(I2).  |p1_j - mean1| <= |p1_j| + |mean1|.  So
//     residual_ref_j < thr   ==>   residual_j < thr + a + b |p1_j| + g |p0_j|,   g = CB eps32, b = g (2 + kappa),
//                                                                                a = g ((2 + kappa) |mean1| + |mean0|)   (|T| <= |mean0| + |mean1|)
// and the number of pairs passing the right-hand test is an upper bound `hi` on the reference's inlier count.
// kind 1 -- the sign of det H is not safe against the float32 rounding of H (|det| <= 64 eps32 sum |cof_ij| A_ij: a sample with
// a repeated point, four coplanar points): the reference's SVD picks the third singular vectors' signs, its pose is one of the
// two of rigid_two_candidates; both are scored and the larger bound is kept.  kind 2 -- rank <= 1: no bound (hi = N).
// Evaluated BEFORE the pose is derived, so that nothing of the sample but (a, b, kind) stays live across the residual pass.
#define RB_G 9.5367431640625e-7f  // CB eps32 = 16 * 2^-24
struct HypBound {
    float a, b;
    int kind;
};
__device__ inline void hypothesis_bound(const Sample4 &s, HypBound &hb) {
    const double *H = s.H;
    const double C0 = H[4] * H[8] - H[5] * H[7], C1 = H[5] * H[6] - H[3] * H[8], C2 = H[3] * H[7] - H[4] * H[6];
    const double C3 = H[2] * H[7] - H[1] * H[8], C4 = H[0] * H[8] - H[2] * H[6], C5 = H[1] * H[6] - H[0] * H[7];
    const double C6 = H[1] * H[5] - H[2] * H[4], C7 = H[2] * H[3] - H[0] * H[5], C8 = H[0] * H[4] - H[1] * H[3];
    const double det = H[0] * C0 + H[1] * C1 + H[2] * C2;
    double I1 = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) I1 += H[i] * H[i];
    const double I2 = C0 * C0 + C1 * C1 + C2 * C2 + C3 * C3 + C4 * C4 + C5 * C5 + C6 * C6 + C7 * C7 + C8 * C8;
    // sum |cof_ij| A_ij, A_ij = sum_q |c1_qi| |c0_qj|: what a relative rounding error of every product can do to det H
    float sumCA = 0.f;
    const float aC[9] = {(float)fabs(C0), (float)fabs(C1), (float)fabs(C2), (float)fabs(C3), (float)fabs(C4), (float)fabs(C5), (float)fabs(C6), (float)fabs(C7), (float)fabs(C8)};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) t += fabsf(s.c1[q][i]) * fabsf(s.c0[q][j]);
            sumCA += aC[3 * i + j] * t;
        }
    hb.kind = 0;
    hb.a = 0.f;
    hb.b = 0.f;
    if (!(I2 > 1e-10 * I1 * I1) || !(I1 > 0.0)) { hb.kind = 2; return; }
    const double kappa = 1.41421356237309515 * I1 / sqrt(I2);
    if (fabs(det) <= 64.0 * 5.9604644775390625e-8 * 1.001 * (double)sumCA) hb.kind = 1;
    // ... nor against the float64 SVD's own backward error (|E| ~ 1e2 eps64 |H|: det moves by <= |cof H|_F |E|_F) when the four points are
    // coplanar to a part in 1e12 -- key points on the ground plane of a mm-quantised scan (round 6, tests/golden/ransac_bound_case.npz)
    if (det * det <= 1e-24 * I1 * I2) hb.kind = 1;   // |det| <= 1e-12 |H|_F |cof H|_F
    const double nm0 = sqrt(s.m0[0] * s.m0[0] + s.m0[1] * s.m0[1] + s.m0[2] * s.m0[2]);
    const double nm1 = sqrt(s.m1[0] * s.m1[0] + s.m1[1] * s.m1[1] + s.m1[2] * s.m1[2]);
    // |T|_inf <= |mean0| + |R mean1| = |mean0| + |mean1| for every candidate pose (T = mean0 - R mean1, Match.py:157)
    // (rounded up: the float conversions and the float evaluation of the test are covered by the 1.0001)
    hb.a = (float)(1.0001 * (double)RB_G * ((2.0 + kappa) * nm1 + nm0));
    hb.b = (float)(1.0001 * (double)RB_G * (2.0 + kappa));
}

// Two launches per set of pairs.
//   k_ransac_hyp     the 500 hypotheses of the first threshold level (0.4 m), four per wavefront, 16 per workgroup:
//                    the matched pairs (P0[pair_idx[i]], P1[i]) are gathered once per workgroup into LDS (coalesced;
//                    the residual loop then never touches global memory), Kabsch on the 4-samples (one hypothesis
//                    per lane group), residuals of the four poses in one pass, ballot + popcount -> counts[trial].
//   k_ransac_finish  one workgroup per pair: replays the sequential accept / early-exit rules over the counts
//                    (Match.py:181-206) as a prefix maximum; if the level failed (no hypothesis reached leastInliers,
//                    :207-214) it evaluates the next level's 500 hypotheses itself, four per wavefront on eight
//                    wavefronts (150 us per level; one per wavefront on four took 1.2 ms, which a pipeline felt) -- up to 1.6 m; then the winner's pose, the inlier mask (:193-194), the inlier count
//                    and the refit over all inliers (:273-282).
// Everything that crosses workgroups crosses a kernel boundary.  Round 1 finished inside the last workgroup of a
// per-level launch (tickets, a `finished` flag, a self-cleaning workspace) and evaluated the mask from the pairs that
// workgroup had staged in LDS; under concurrent streams ~1 call in 3 000 then stored 64-element runs of constant 0 / 1
// from some of its wavefronts although the inlier COUNT summed from the same registers was right (tools/stress_mask.py:
// sentinel-filled buffer, D2H cross-check, exactly one replay per call counted) -- the "3 of 41 suite runs" of round 1.
// The hand-off protocol was not involved.  This structure shows 0 mismatches in 72 000 frames.
#define RE_WAVES 4
#define RE_LDS_PAIRS 1024

// Every lane of a wavefront derives the same hypothesis from the same four pairs; lanes that disagree mean the hardware
// mis-executed something (with packed-f32 instructions enabled, gfx950 sporadically drops the subtrahend of
// one `v_pk_add_f32 ... op_sel:[0,1]` in lanes 48-63 when three queues are busy -- the library is built without them).
// The check costs a dozen v_readfirstlane per hypothesis and turns a silent wrong inlier count into a counter that
// caelo_lane_faults() reports and the tests and bench.py assert to be zero.
__device__ inline void lane_agreement(const float R[9], const float T[3], int lane, int32_t *faults) {
    if (!faults) return;
    unsigned int diff = 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) diff |= __float_as_uint(R[q]) ^ (unsigned int)__builtin_amdgcn_readfirstlane((int)__float_as_uint(R[q]));
#pragma unroll
    for (int q = 0; q < 3; ++q) diff |= __float_as_uint(T[q]) ^ (unsigned int)__builtin_amdgcn_readfirstlane((int)__float_as_uint(T[q]));
    if (__ballot(diff != 0) != 0ull && lane == 0) atomicAdd(faults, 1);
}

// inlier count of one hypothesis by one wavefront (every lane returns it)
__device__ inline int hypothesis_count(const float *P0, int l0, const int64_t *pidx, const float *P1, int l1, int N, const double *r4,
                                       float thr, int lane, int32_t *faults) {
    float R[9], T[3];
    sample_hypothesis(P0, l0, pidx, P1, l1, N, r4, R, T);
    lane_agreement(R, T, lane, faults);
    int cnt = 0;  // residuals + inlier count (:191-194): ballot + popcount per 64 pairs
    for (int i = lane; i < ((N + 63) & ~63); i += 64) {
        bool in = false;
        if (i < N) {
            const float *a = P0 + (size_t)l0 * (pidx ? pidx[i] : i);
            const float *b = P1 + (size_t)l1 * i;
            in = residual(R, T, a[0], a[1], a[2], b[0], b[1], b[2]) < thr;
        }
        cnt += __popcll(__ballot(in));
    }
    return cnt;
}

// One wavefront = RH_PER_WAVE hypotheses: lane l derives hypothesis l & 3 (the Kabsch step is a long serial f64 chain
// that costs the same for 1 or 64 lanes; the 16 lanes that share a hypothesis must agree -- the hardware self-check),
// then the four poses are broadcast through scalar registers and ONE pass over the staged pairs counts the inliers of
// all four (a pair is read from LDS once).  Round 2 before: one hypothesis per wavefront, 35 us per 8 pairs.
// `rnd`: the draws of the level (trial t at rnd + 4 t); counts[trial0 .. trial0 + 3] are written.
#define RH_PER_WAVE 4
// CERT: also the certificate's bound (sN: per pair (|p1_j|, g |p0_j|); cert: the record `hi` / `idx` of the trials go to)
template <bool CERT>
__device__ __forceinline__ void four_hypotheses(const float *sP0, const float *sP1, int N, const double *rnd, int trial0, float thr, int lane,
                                       int32_t *faults, int32_t *counts, const float2 *sN = nullptr, caelo_ransac_cert *cert = nullptr,
                                       int level = 0) {
    // (__forceinline__: with a second caller -- k_ransac_hyp_up -- hipcc stopped inlining this function, and a CALL costs the callers
    // the full register budget: k_ransac_hyp went from 168 to 248 registers, no longer fitted beside two stage-1 workgroups, and its
    // launch stretched from 74 to 199 us inside the pipeline, profiles/r06_kernel_stats_bench.txt)
    // ---- lane l: hypothesis trial0 + (l & 3) (a trial past the last repeats the last one; its count is not stored)
    const int mine = min(trial0 + (lane & (RH_PER_WAVE - 1)), CAELO_RANSAC_MAX_TRIALS - 1);
    float R[9], T[3];
    HypBound hb = {0.f, 0.f, 0};
    if (CERT) {
        Sample4 smp;
        sample4(sP0, 3, nullptr, sP1, 3, N, rnd + (size_t)mine * 4, smp);
        hypothesis_bound(smp, hb);
        if (lane < RH_PER_WAVE && trial0 + lane < CAELO_RANSAC_MAX_TRIALS)
#pragma unroll
            for (int q = 0; q < 4; ++q) (level ? cert->idx_up[level - 1] : cert->idx)[trial0 + lane][q] = smp.idx[q];
        rigid_from_H<false>(smp.H, smp.m0, smp.m1, R, T);   // (a rank-2 sample is kind 1: both poses are scored below)
    } else {
        sample_hypothesis(sP0, 3, nullptr, sP1, 3, N, rnd + (size_t)mine * 4, R, T);
    }
    if (faults) {  // the 16 lanes of a hypothesis hold the same pose, bit for bit
        unsigned int hsh = 0;
#pragma unroll
        for (int q = 0; q < 9; ++q) hsh = hsh * 0x9E3779B1u + __float_as_uint(R[q]);
#pragma unroll
        for (int q = 0; q < 3; ++q) hsh = hsh * 0x9E3779B1u + __float_as_uint(T[q]);
        const bool bad = hsh != (unsigned int)__shfl_xor((int)hsh, 4) || hsh != (unsigned int)__shfl_xor((int)hsh, 16) ||
                         hsh != (unsigned int)__shfl_xor((int)hsh, 32);
        if (__ballot(bad) != 0ull && lane == 0) atomicAdd(faults, 1);
    }
    // ---- the four poses in scalar registers, one pass over the pairs
    float Rs[RH_PER_WAVE][9], Ts[RH_PER_WAVE][3];
#pragma unroll
    for (int h = 0; h < RH_PER_WAVE; ++h) {
#pragma unroll
        for (int q = 0; q < 9; ++q) Rs[h][q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(R[q]), h));
#pragma unroll
        for (int q = 0; q < 3; ++q) Ts[h][q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(T[q]), h));
    }
    int cnt[RH_PER_WAVE] = {0, 0, 0, 0};
    if (!CERT) {
        for (int i = lane; i < ((N + 63) & ~63); i += 64) {
            const bool live = i < N;
            const int ii = live ? i : 0;
            const float ax = sP0[3 * ii], ay = sP0[3 * ii + 1], az = sP0[3 * ii + 2];
            const float bx = sP1[3 * ii], by = sP1[3 * ii + 1], bz = sP1[3 * ii + 2];
#pragma unroll
            for (int h = 0; h < RH_PER_WAVE; ++h) {
                const bool in = live && residual(Rs[h], Ts[h], ax, ay, az, bx, by, bz) < thr;
                cnt[h] += __popcll(__ballot(in));
            }
        }
    } else {
        // ---- the same pass with the certificate's second test (hypothesis_bound); rank-2 samples get their two candidate poses
        // scored as well, rank-1 samples no bound
        float as[RH_PER_WAVE], bs[RH_PER_WAVE];
        int kinds[RH_PER_WAVE];
#pragma unroll
        for (int h = 0; h < RH_PER_WAVE; ++h) {
            as[h] = thr + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hb.a), h));
            bs[h] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hb.b), h));
            kinds[h] = __builtin_amdgcn_readlane(hb.kind, h);
        }
        int hi[RH_PER_WAVE] = {0, 0, 0, 0};
        for (int i = lane; i < ((N + 63) & ~63); i += 64) {
            const bool live = i < N;
            const int ii = live ? i : 0;
            const float ax = sP0[3 * ii], ay = sP0[3 * ii + 1], az = sP0[3 * ii + 2];
            const float bx = sP1[3 * ii], by = sP1[3 * ii + 1], bz = sP1[3 * ii + 2];
            const float2 nn = sN[ii];
#pragma unroll
            for (int h = 0; h < RH_PER_WAVE; ++h) {
                const float r = residual(Rs[h], Ts[h], ax, ay, az, bx, by, bz);
                cnt[h] += __popcll(__ballot(live && r < thr));
                hi[h] += __popcll(__ballot(live && r < as[h] + fmaf(bs[h], nn.x, nn.y)));
            }
        }
        if ((kinds[0] | kinds[1] | kinds[2] | kinds[3]) & 1) {  // wave-uniform: some hypothesis of this wavefront has two candidate poses
            // (derived again from the sample: cheaper than carrying 24 more registers through the pass every wavefront runs)
            float Ra[9], Ta[3], Rb[9], Tb[3];
            bool two;
            {
                Sample4 smp;
                sample4(sP0, 3, nullptr, sP1, 3, N, rnd + (size_t)mine * 4, smp);
                two = rigid_two_candidates(smp.H, smp.m0, smp.m1, Ra, Ta, Rb, Tb);
            }
            int ha[RH_PER_WAVE] = {0, 0, 0, 0};
#pragma unroll 1
            for (int h = 0; h < RH_PER_WAVE; ++h) {
                if (kinds[h] != 1) continue;  // (scalar condition)
                if (!__builtin_amdgcn_readlane((int)two, h)) { ha[h] = N; continue; }
                float Ras[9], Tas[3], Rbs[9], Tbs[3];
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    Ras[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Ra[q]), h));
                    Rbs[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Rb[q]), h));
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    Tas[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Ta[q]), h));
                    Tbs[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Tb[q]), h));
                }
                int na = 0, nb = 0;
                for (int i = lane; i < ((N + 63) & ~63); i += 64) {
                    const bool live = i < N;
                    const int ii = live ? i : 0;
                    const float ax = sP0[3 * ii], ay = sP0[3 * ii + 1], az = sP0[3 * ii + 2];
                    const float bx = sP1[3 * ii], by = sP1[3 * ii + 1], bz = sP1[3 * ii + 2];
                    const float2 nn = sN[ii];
                    const float lim = as[h] + fmaf(bs[h], nn.x, nn.y);
                    na += __popcll(__ballot(live && residual(Ras, Tas, ax, ay, az, bx, by, bz) < lim));
                    nb += __popcll(__ballot(live && residual(Rbs, Tbs, ax, ay, az, bx, by, bz) < lim));
                }
                ha[h] = max(na, nb);
            }
#pragma unroll
            for (int h = 0; h < RH_PER_WAVE; ++h)
                if (kinds[h] == 1) hi[h] = max(hi[h], ha[h]);
        }
#pragma unroll
        for (int h = 0; h < RH_PER_WAVE; ++h)
            if (kinds[h] == 2) hi[h] = N;
        if (lane < RH_PER_WAVE && trial0 + lane < CAELO_RANSAC_MAX_TRIALS)
            (level ? cert->hi_up[level - 1] : cert->hi)[trial0 + lane] = lane == 0 ? hi[0] : (lane == 1 ? hi[1] : (lane == 2 ? hi[2] : hi[3]));
    }
    if (lane < RH_PER_WAVE && trial0 + lane < CAELO_RANSAC_MAX_TRIALS)
        counts[trial0 + lane] = lane == 0 ? cnt[0] : (lane == 1 ? cnt[1] : (lane == 2 ? cnt[2] : cnt[3]));
}

__global__ void __launch_bounds__(64 * RE_WAVES) k_ransac_hyp(const caelo_pair_set ps, int ld0, int ld1, int64_t k1_max) {
    const caelo_pair_dev &P = ps.p[blockIdx.z];
    const float *__restrict__ pc0 = P.pc0, *__restrict__ pc1 = P.pc1;
    const int64_t *__restrict__ pair_idx = P.pair_idx;
    RansacWs *ws = (RansacWs *)P.ws_ransac;
    __shared__ float sP0[RE_LDS_PAIRS * 3], sP1[RE_LDS_PAIRS * 3];
    __shared__ float2 sN[RE_LDS_PAIRS];  // certificate: (|p1_j|, g |p0_j|) of hypothesis_bound
    caelo_ransac_cert *cert = P.cert;
    // no pairs at all when EITHER frame has no key point (frame 0 empty: the match kernel wrote index 0 everywhere, the
    // reference's argmin over an empty axis raises): the pose fails as a value
    const int N = (P.n0 && *P.n0 <= 0) ? 0 : (P.n1 ? min(max(*P.n1, 0), (int)k1_max) : (int)k1_max);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // pairs -> LDS (falls back to the global arrays when they do not fit)
    const bool in_lds = N <= RE_LDS_PAIRS;
    if (in_lds) {
        for (int i = tid; i < N; i += 64 * RE_WAVES) {
            const float *a = pc0 + (size_t)ld0 * pair_idx[i];
            const float *b = pc1 + (size_t)ld1 * i;
            const float a0 = a[0], a1 = a[1], a2 = a[2], b0 = b[0], b1 = b[1], b2 = b[2];
            sP0[3 * i] = a0; sP0[3 * i + 1] = a1; sP0[3 * i + 2] = a2;
            sP1[3 * i] = b0; sP1[3 * i + 1] = b1; sP1[3 * i + 2] = b2;
            if (cert) sN[i] = make_float2(1.0001f * sqrtf(b0 * b0 + b1 * b1 + b2 * b2), 1.0001f * RB_G * sqrtf(a0 * a0 + a1 * a1 + a2 * a2));
        }
    }
    __syncthreads();
    if (cert && blockIdx.x == 0 && tid == 0) cert->levels_up = 0;   // (k_ransac_hyp_up, the next launch, sets it when it writes the higher levels' bounds)
    if (cert && P.cert_only && blockIdx.x == 0) {
        // the host half takes the pair from here (caelo_pipeline with result_host): the certificate's pairs and header leave with
        // this launch and k_ransac_finish is not launched at all -- what it computes (the kernels' own winner, mask and refit) is
        // exactly what the host half replaces
        if (in_lds) {
            float *d0 = &cert->p0[0][0], *d1 = &cert->p1[0][0];
            for (int i = tid; i < 3 * N; i += 64 * RE_WAVES) { d0[i] = sP0[i]; d1[i] = sP1[i]; }
        }
        if (tid == 0) {
            cert->n_pairs = N;
            cert->flags = in_lds ? 0 : CAELO_CERT_NO_BOUNDS;
            cert->magic = CAELO_CERT_MAGIC;
        }
    }
    const int trial0 = (blockIdx.x * RE_WAVES + wave) * RH_PER_WAVE;  // wave-uniform
    if (trial0 >= CAELO_RANSAC_MAX_TRIALS) return;
    if (!in_lds) {  // more pairs than the LDS stage holds: one hypothesis at a time from global memory
        for (int h = 0; h < RH_PER_WAVE && trial0 + h < CAELO_RANSAC_MAX_TRIALS; ++h) {
            const int cnt = hypothesis_count(pc0, ld0, pair_idx, pc1, ld1, N, P.rand + (size_t)(trial0 + h) * 4, 0.4f, lane, ps.faults);
            if (lane == 0) ws->counts[trial0 + h] = cnt;
        }
        return;
    }
    if (cert) four_hypotheses<true>(sP0, sP1, N, P.rand, trial0, 0.4f, lane, ps.faults, ws->counts, sN, cert, 0);
    else four_hypotheses<false>(sP0, sP1, N, P.rand, trial0, 0.4f, lane, ps.faults, ws->counts);
}

// Round 6: certificates for the 0.8 m and 1.6 m levels (Match.py:207-214).  A pair escalates when no hypothesis of a level reaches
// leastInliers; without bounds for the next level the host half evaluates all of its (up to 500) hypotheses like the reference's loop --
// 2.7 ms of a host core per level, and on data with real failures the exact path becomes a CPU path (missing 3).  Launched
// behind k_ransac_hyp (before k_ransac_finish reuses the counts) for pairs that leave a certificate: a workgroup first looks at the first level's 500 counts and
// returns unless none reached leastInliers (the device's own float64 counts: the reference's may differ by a pair at the threshold,
// in which case the host half meets a level without bounds and evaluates it the long way -- still exact); otherwise blockIdx.y's level
// (0.8 m, 1.6 m: both, the second is wasted when the first succeeds, and escalations are rare) gets its `hi_up` / `idx_up` exactly as
// the first level got `hi` / `idx`.  The inlier counts of these levels are not kept (scratch): the host half derives them.
#define RU_BLOCKS 32   // (8 workgroups per level were tried for the sake of the launches that only look and leave: those took as long as before, 43 us on the pair stream, and a batch with a failing pair 4 x longer -- the failing_pairs leg fell from 17.5 k to 9.9 k frames/s)
__global__ void __launch_bounds__(64 * RE_WAVES) k_ransac_hyp_up(const caelo_pair_set ps, int ld0, int ld1, int64_t k1_max) {
    const caelo_pair_dev &P = ps.p[blockIdx.z];
    caelo_ransac_cert *cert = P.cert;
    if (!cert) return;
    RansacWs *ws = (RansacWs *)P.ws_ransac;
    const int N = (P.n0 && *P.n0 <= 0) ? 0 : (P.n1 ? min(max(*P.n1, 0), (int)k1_max) : (int)k1_max);
    if (N > RE_LDS_PAIRS) return;
    const int least = min(100, (int)(0.2 * (double)N));   // :166
    if (least <= 0) return;                                // N < 5: the first level "succeeds" without any inlier (:195-205)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bool reached = false;
    for (int t = tid; t < CAELO_RANSAC_MAX_TRIALS; t += 64 * RE_WAVES) reached = reached || ws->counts[t] >= least;
    if (__syncthreads_or(reached)) return;
    __shared__ float sP0[RE_LDS_PAIRS * 3], sP1[RE_LDS_PAIRS * 3];
    __shared__ float2 sN[RE_LDS_PAIRS];
    __shared__ int32_t s_scratch[CAELO_RANSAC_MAX_TRIALS];
    const float *__restrict__ pc0 = P.pc0, *__restrict__ pc1 = P.pc1;
    const int64_t *__restrict__ pair_idx = P.pair_idx;
    for (int i = tid; i < N; i += 64 * RE_WAVES) {
        const float *a = pc0 + (size_t)ld0 * pair_idx[i];
        const float *b = pc1 + (size_t)ld1 * i;
        const float a0 = a[0], a1 = a[1], a2 = a[2], b0 = b[0], b1 = b[1], b2 = b[2];
        sP0[3 * i] = a0; sP0[3 * i + 1] = a1; sP0[3 * i + 2] = a2;
        sP1[3 * i] = b0; sP1[3 * i + 1] = b1; sP1[3 * i + 2] = b2;
        sN[i] = make_float2(1.0001f * sqrtf(b0 * b0 + b1 * b1 + b2 * b2), 1.0001f * RB_G * sqrtf(a0 * a0 + a1 * a1 + a2 * a2));
    }
    __syncthreads();
    const int level = 1 + (int)blockIdx.y;   // 1, 2
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) cert->levels_up = 1;   // (k_ransac_hyp wrote 0 with the header)
    // (RU_BLOCKS workgroups per level and pair, each walking its share of the 500 trials: 512 workgroups of 33 KB of LDS that only
    // look at 500 counts and leave took 48 us of the pair stream per batch to be scheduled beside the encoder's persistent grids)
    for (int trial0 = (blockIdx.x * RE_WAVES + wave) * RH_PER_WAVE; trial0 < CAELO_RANSAC_MAX_TRIALS; trial0 += RU_BLOCKS * RE_WAVES * RH_PER_WAVE)
        four_hypotheses<true>(sP0, sP1, N, P.rand + (size_t)level * CAELO_RANSAC_MAX_TRIALS * 4, trial0, 0.4f * (float)(1 << level), lane, ps.faults,
                              s_scratch, sN, cert, level);
}

#define RF_WAVES 8   // the accept rules, the mask and the refit use four of them; all eight evaluate a next level's hypotheses
__global__ void __launch_bounds__(64 * RF_WAVES, 4) k_ransac_finish(const caelo_pair_set ps, int ld0, int ld1, int64_t k1_max) {
    const caelo_pair_dev &P = ps.p[blockIdx.z];
    if (P.cert && P.cert_only) return;   // (a mixed set: this pair's result comes from the host half)
    const float *__restrict__ pc0 = P.pc0, *__restrict__ pc1 = P.pc1;
    const int64_t *__restrict__ pair_idx = P.pair_idx;
    const double *__restrict__ rnd = P.rand;
    RansacWs *ws = (RansacWs *)P.ws_ransac;
    caelo_pose_result *res = P.result;
    uint8_t *mask = P.mask;
    __shared__ float sP0[RE_LDS_PAIRS * 3], sP1[RE_LDS_PAIRS * 3];
    __shared__ float Rs[9], Ts[3];
    __shared__ RansacVerdict s_v;
    __shared__ double red[4][FIT_TERMS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // no pairs at all when EITHER frame has no key point (frame 0 empty: the match kernel wrote index 0 everywhere, the
    // reference's argmin over an empty axis raises): the pose fails as a value
    const int N = (P.n0 && *P.n0 <= 0) ? 0 : (P.n1 ? min(max(*P.n1, 0), (int)k1_max) : (int)k1_max);
    // ---- the first level's counts (wavefront 0, eight per lane) are fetched while all four wavefronts gather the matched
    // pairs into LDS once: the winner's sample, the inlier mask and the refit below then never touch global memory again
    // (round 2 before: three passes of dependent global gathers in this one-workgroup kernel, 27 us)
    int c0[(CAELO_RANSAC_MAX_TRIALS + 63) / 64];
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < (CAELO_RANSAC_MAX_TRIALS + 63) / 64; ++q) {
            const int i = lane * ((CAELO_RANSAC_MAX_TRIALS + 63) / 64) + q;
            c0[q] = i < CAELO_RANSAC_MAX_TRIALS ? ws->counts[i] : 0;
        }
    }
    const bool in_lds = N <= RE_LDS_PAIRS;
    if (in_lds) {
        for (int i = tid; i < N; i += 64 * RF_WAVES) {
            const float *a = pc0 + (size_t)ld0 * pair_idx[i];
            const float *b = pc1 + (size_t)ld1 * i;
            sP0[3 * i] = a[0]; sP0[3 * i + 1] = a[1]; sP0[3 * i + 2] = a[2];
            sP1[3 * i] = b[0]; sP1[3 * i + 1] = b[1]; sP1[3 * i + 2] = b[2];
        }
    }
    // ---- accept rules, level by level (:207-214: thr doubles while no hypothesis reaches leastInliers, up to 1.6)
    if (tid < 64) ransac_replay(N, ws->counts, &s_v, c0);
    __syncthreads();  // also the end of the staging above
    if (P.cert) {  // the certificate's header and the matched pairs (k_ransac_hyp wrote `hi` and `idx`)
        caelo_ransac_cert *cert = P.cert;
        if (in_lds) {
            float *d0 = &cert->p0[0][0], *d1 = &cert->p1[0][0];
            for (int i = tid; i < 3 * N; i += 64 * RF_WAVES) { d0[i] = sP0[i]; d1[i] = sP1[i]; }
        }
        if (tid == 0) {
            cert->n_pairs = N;
            cert->flags = in_lds ? 0 : CAELO_CERT_NO_BOUNDS;
            cert->magic = CAELO_CERT_MAGIC;
        }
    }
    int level = 0;
    while (!s_v.success && level < CAELO_RANSAC_LEVELS - 1) {  // rare: the next level's 500 hypotheses, four per wavefront like k_ransac_hyp
        __syncthreads();  // everyone has read s_v before it is overwritten
        ++level;
        const float thr_l = 0.4f * (float)(1 << level);
        const double *rnd_l = rnd + (size_t)level * CAELO_RANSAC_MAX_TRIALS * 4;
        if (in_lds) {
            for (int trial0 = wave * RH_PER_WAVE; trial0 < CAELO_RANSAC_MAX_TRIALS; trial0 += RF_WAVES * RH_PER_WAVE)
                four_hypotheses<false>(sP0, sP1, N, rnd_l, trial0, thr_l, lane, ps.faults, ws->counts);
        } else {
            for (int trial = wave; trial < CAELO_RANSAC_MAX_TRIALS; trial += RF_WAVES) {
                const int cnt = hypothesis_count(pc0, ld0, pair_idx, pc1, ld1, N, rnd_l + (size_t)trial * 4, thr_l, lane, ps.faults);
                if (lane == 0) ws->counts[trial] = cnt;
            }
        }
        __threadfence_block();
        __syncthreads();
        if (tid < 64) ransac_replay(N, ws->counts, &s_v);
        __syncthreads();
    }
    const int success = s_v.success, best = s_v.best_trial;
    const float thr = 0.4f * (float)(1 << level);  // 0.4, 0.8, 1.6 (:171,:210)
    // ---- the winner is recomputed from its sample (deterministic), identity if every level failed (:177)
    if (tid < 64) {
        float R[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}, T[3] = {0.f, 0.f, 0.f};
        if (best >= 0) {
            const double *r4 = rnd + ((size_t)level * CAELO_RANSAC_MAX_TRIALS + best) * 4;
            if (in_lds) sample_hypothesis(sP0, 3, nullptr, sP1, 3, N, r4, R, T);
            else sample_hypothesis(pc0, ld0, pair_idx, pc1, ld1, N, r4, R, T);
        }
        lane_agreement(R, T, lane, ps.faults);
        if (tid < 9) Rs[tid] = R[tid];
        if (tid < 3) Ts[tid] = T[tid];
    }
    __syncthreads();
    // ---- one pass: inlier mask (four consecutive pairs per thread, one aligned 32-bit store: the whole mask row goes out as full
    // dwords) and the sums of the refit over the inliers (:193-194, :273-282)
    double acc[FIT_TERMS];
#pragma unroll
    for (int t = 0; t < FIT_TERMS; ++t) acc[t] = 0.0;
    const bool word_ok = (((uintptr_t)mask) & 3u) == 0;
    for (int i0 = tid * 4; tid < 256 && i0 < (int)k1_max; i0 += 4 * 256) {   // (the first four wavefronts, as before there were eight)
        unsigned int packed = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q;
            if (i < N && best >= 0) {
                const float *a = in_lds ? sP0 + 3 * i : pc0 + (size_t)ld0 * pair_idx[i];
                const float *b = in_lds ? sP1 + 3 * i : pc1 + (size_t)ld1 * i;
                const float ax = a[0], ay = a[1], az = a[2], bx = b[0], by = b[1], bz = b[2];
                if (residual(Rs, Ts, ax, ay, az, bx, by, bz) < thr) {
                    packed |= 1u << (8 * q);
                    const double x0 = ax, y0 = ay, z0 = az, x1 = bx, y1 = by, z1 = bz;
                    acc[0] += 1.0;
                    acc[1] += x0; acc[2] += y0; acc[3] += z0;
                    acc[4] += x1; acc[5] += y1; acc[6] += z1;
                    acc[7] += x1 * x0; acc[8] += x1 * y0; acc[9] += x1 * z0;   // P1^T P0  (:146)
                    acc[10] += y1 * x0; acc[11] += y1 * y0; acc[12] += y1 * z0;
                    acc[13] += z1 * x0; acc[14] += z1 * y0; acc[15] += z1 * z0;
                }
            }
        }
        if (word_ok && i0 + 3 < (int)k1_max) *(unsigned int *)(mask + i0) = packed;
        else
            for (int q = 0; q < 4 && i0 + q < (int)k1_max; ++q) mask[i0 + q] = (uint8_t)((packed >> (8 * q)) & 1u);
    }
#pragma unroll
    for (int t = 0; t < FIT_TERMS; ++t)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[t] += __shfl_xor(acc[t], o);
    if (lane == 0 && wave < 4)
#pragma unroll
        for (int t = 0; t < FIT_TERMS; ++t) red[wave][t] = acc[t];
    __syncthreads();
    if (tid < 9) { res->R_ransac[tid] = Rs[tid]; }
    if (tid < 3) { res->T_ransac[tid] = Ts[tid]; }
    if (tid == 0) {
        double sm[FIT_TERMS];
#pragma unroll
        for (int t = 0; t < FIT_TERMS; ++t) sm[t] = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
        const int n_in = (int)sm[0];
        res->threshold = thr;
        res->success = success;
        res->iterations = s_v.iterations;
        res->n_inliers = n_in;
        res->best_trial = best >= 0 ? level * CAELO_RANSAC_MAX_TRIALS + best : -1;
        res->n_pairs = N;
        float R[9], T[3];
#pragma unroll
        for (int q = 0; q < 9; ++q) R[q] = Rs[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) T[q] = Ts[q];
        if (n_in > 0) {  // :277-282: the pose over all inliers
            const double cnt = sm[0];
            const double m0[3] = {sm[1] / cnt, sm[2] / cnt, sm[3] / cnt}, m1[3] = {sm[4] / cnt, sm[5] / cnt, sm[6] / cnt};
            double H[9];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) H[3 * i + j] = sm[7 + 3 * i + j] - cnt * m1[i] * m0[j];
            rigid_from_H(H, m0, m1, R, T);
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) res->R[q] = R[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) res->T[q] = T[q];
    }
}

CAELO_API int caelo_ransac(caelo_ctx *c, const float *pc0, int ld0, const float *pc1, int ld1, const int64_t *pair_idx,
                           int64_t k1_max, const int32_t *n1, const double *rnd, caelo_pose_result *result,
                           uint8_t *inlier_mask, void *wsv, caelo_ransac_cert *cert, void *stream) {
    CAELO_REQUIRE(c && pc0 && pc1 && pair_idx && rnd && result && inlier_mask && wsv, "null argument");
    CAELO_REQUIRE((((uintptr_t)cert) & 15u) == 0, "certificate not 16-byte aligned");
    caelo_pair_set ps = {};
    ps.n = 1;
    ps.faults = c->faults;
    caelo_pair_dev &p = ps.p[0];
    p.pc0 = pc0; p.pc1 = pc1; p.pair_idx = const_cast<int64_t *>(pair_idx); p.n1 = n1; p.rand = rnd; p.result = result;
    p.mask = inlier_mask; p.ws_ransac = wsv; p.cert = cert;
    return ransac_set(ps, ld0, ld1, k1_max, caelo_stream(stream));
}

int ransac_set(const caelo_pair_set &ps, int ld0, int ld1, int64_t k1_max, hipStream_t s) {
    CAELO_REQUIRE(ps.n >= 1 && ps.n <= CAELO_FB_MAX, "bad pair count");
    CAELO_REQUIRE(k1_max > 0 && ld0 >= 3 && ld1 >= 3, "bad shape");
    k_ransac_hyp<<<dim3((CAELO_RANSAC_MAX_TRIALS + RE_WAVES * RH_PER_WAVE - 1) / (RE_WAVES * RH_PER_WAVE), 1, ps.n), 64 * RE_WAVES, 0, s>>>(ps, ld0, ld1, k1_max);
    CAELO_LAUNCH_CHECK();
    bool all_cert_only = true;
    for (int i = 0; i < ps.n; ++i) all_cert_only = all_cert_only && ps.p[i].cert && ps.p[i].cert_only;
    bool any_cert = false;
    for (int i = 0; i < ps.n; ++i) any_cert = any_cert || ps.p[i].cert != nullptr;
    if (any_cert) {   // bounds for the 0.8 / 1.6 m levels of the pairs whose first level failed (workgroups of the others return at once)
        k_ransac_hyp_up<<<dim3(RU_BLOCKS, 2, ps.n), 64 * RE_WAVES, 0, s>>>(ps, ld0, ld1, k1_max);
        CAELO_LAUNCH_CHECK();
    }
    if (all_cert_only) return CAELO_OK;   // the host half decides every pair of the set: no finishing kernel
    k_ransac_finish<<<dim3(1, 1, ps.n), 64 * RF_WAVES, 0, s>>>(ps, ld0, ld1, k1_max);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

// ------------------------------------------------------------------------------------------------
// ICP step: MyICP.py:26-72 / :75-85 -- the refinement that follows the odometry
// ------------------------------------------------------------------------------------------------
// One iteration of the reference's point-to-point ICP: for every point of PC1 its nearest neighbour in PC0
// (sklearn NearestNeighbors(n_neighbors=1): exact Euclidean distance in float64), the pairs closer than the
// threshold, SolveRT on them, PC1 <- R PC1 + T.  The iteration control (threshold decay, Euler-angle stop rule)
// stays on the host, like the reference's Python loop.
#define ICP_TILE 1024

__global__ void __launch_bounds__(256) k_icp_nn(const float *__restrict__ pc0, int n0, const float *__restrict__ pc1, int n1,
                                                double thr, int64_t *__restrict__ idx0, uint8_t *__restrict__ mask,
                                                int32_t *__restrict__ n_in) {
    __shared__ float tile[ICP_TILE * 3];
    __shared__ int s_cnt;
    const int tid = threadIdx.x;
    const int j = blockIdx.x * blockDim.x + tid;
    if (tid == 0) s_cnt = 0;
    double qx = 0.0, qy = 0.0, qz = 0.0;
    if (j < n1) { qx = pc1[3 * (size_t)j]; qy = pc1[3 * (size_t)j + 1]; qz = pc1[3 * (size_t)j + 2]; }
    double best = 1.0e300;
    int besti = 0;
    for (int base = 0; base < n0; base += ICP_TILE) {
        const int m = min(ICP_TILE, n0 - base);
        __syncthreads();
        for (int i = tid; i < 3 * m; i += 256) tile[i] = pc0[3 * (size_t)base + i];
        __syncthreads();
        for (int i = 0; i < m; ++i) {  // all lanes read the same address: LDS broadcast
            const double dx = (double)tile[3 * i] - qx, dy = (double)tile[3 * i + 1] - qy, dz = (double)tile[3 * i + 2] - qz;
            const double d2 = dx * dx + dy * dy + dz * dz;  // sequential x, y, z like the kd-tree's reduced distance
            if (d2 < best) { best = d2; besti = base + i; }
        }
    }
    const bool in = j < n1 && sqrt(best) < thr;  // GetPtsInliners: distances < inlierThreshold (:80)
    if (j < n1) { idx0[j] = besti; mask[j] = in ? 1 : 0; }
    const unsigned long long bal = __ballot(in);
    if ((tid & 63) == 0 && bal) atomicAdd(&s_cnt, __popcll(bal));
    __syncthreads();
    if (tid == 0 && s_cnt) atomicAdd(n_in, s_cnt);
}

// SolveRT on the inlier pairs and PC1 <- R PC1 + T (one workgroup; float32 like the reference's arrays)
__global__ void __launch_bounds__(256) k_icp_fit_apply(const float *__restrict__ pc0, float *__restrict__ pc1, int n1,
                                                       const int64_t *__restrict__ idx0, const uint8_t *__restrict__ mask,
                                                       const int32_t *__restrict__ n_in, int min_inliers, float *__restrict__ rt) {
    if (*n_in < min_inliers) return;  // the host stops the iteration (MyICP.py:38-40)
    fit_block(pc0, 3, idx0, pc1, 3, mask, n1, rt, rt + 9, nullptr);
    __syncthreads();
    const float r0 = rt[0], r1 = rt[1], r2 = rt[2], r3 = rt[3], r4 = rt[4], r5 = rt[5], r6 = rt[6], r7 = rt[7], r8 = rt[8];
    const float t0 = rt[9], t1 = rt[10], t2 = rt[11];
    for (int j = threadIdx.x; j < n1; j += 256) {  // PC1 = (np.dot(R, PC1.T) + T).T  (:50)
        const float x = pc1[3 * (size_t)j], y = pc1[3 * (size_t)j + 1], z = pc1[3 * (size_t)j + 2];
        pc1[3 * (size_t)j] = r0 * x + r1 * y + r2 * z + t0;
        pc1[3 * (size_t)j + 1] = r3 * x + r4 * y + r5 * z + t1;
        pc1[3 * (size_t)j + 2] = r6 * x + r7 * y + r8 * z + t2;
    }
}

CAELO_API int64_t caelo_icp_ws_bytes(int64_t n1) { return ((n1 * 9 + 255) / 256) * 256 + 256; }

CAELO_API int caelo_icp_step(caelo_ctx *c, const float *pc0, int64_t n0, float *pc1, int64_t n1, double threshold,
                             int min_inliers, float *rt, int32_t *n_inliers, void *ws, void *stream) {
    CAELO_REQUIRE(c && pc0 && pc1 && rt && n_inliers && ws, "null argument");
    CAELO_REQUIRE(n0 > 0 && n1 > 0 && n0 < (1 << 30) && n1 < (1 << 30), "bad shape");
    hipStream_t s = caelo_stream(stream);
    int64_t *idx0 = (int64_t *)ws;
    uint8_t *mask = (uint8_t *)(idx0 + n1);
    CAELO_HIP(hipMemsetAsync(n_inliers, 0, sizeof(int32_t), s));
    k_icp_nn<<<(unsigned)((n1 + 255) / 256), 256, 0, s>>>(pc0, (int)n0, pc1, (int)n1, threshold, idx0, mask, n_inliers);
    CAELO_LAUNCH_CHECK();
    k_icp_fit_apply<<<1, 256, 0, s>>>(pc0, pc1, (int)n1, idx0, mask, n_inliers, min_inliers, rt);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}
