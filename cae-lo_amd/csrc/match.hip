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

#include "caelo_internal.h"

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
// Grid = (column tiles of 16 frame-1 descriptors) x MM_RS row slices; a workgroup's 16 waves take one
// 16-row tile of frame 0 each (more slices of 256 rows when k0 > 1024).  Every workgroup reduces its
// rows to the three smallest lower bounds per column and publishes them; the last slice to arrive (agent
// -scope release / ticket / acquire) merges the slices and certifies.  If a third bound still passes the
// test the column is re-scanned exactly by that workgroup.
// ------------------------------------------------------------------------------------------------
typedef double mm_f64x4 __attribute__((ext_vector_type(4)));
#define MM_WAVES 16
#define MM_KSTEPS 16  // dim <= 64
#define MM_RS 4       // row slices per column tile

struct MmPartial {
    double L1, L2, L3, U;
    int I1, I2;
};
// Cross-workgroup hand-off: the partial results go out as agent-scope atomic stores (write-through to the coherence
// point, sc1) and come back as agent-scope atomic loads (cdna_hip_programming.md G16, "8-B agent atomics both sides");
// the publishing wave waits for its stores to complete (s_waitcnt vmcnt(0)) before the workgroup takes its ticket, an
// agent-scope read-modify-write.  Every shared location is touched by agent-scope atomics only, so no access can be
// served from a non-coherent cache.  CAELO_XWG_FENCES=1 builds the formal release / acquire version instead
// (__threadfence() around an ACQ_REL ticket): on gfx950 an agent-scope release is an L2 write-back walk and an acquire an
// invalidate, PER WORKGROUP -- measured 18.7 -> 33 us per 1024 x 1024 match for bit-identical results
// (tests/test_gpu_parity.py::test_match_ransac_and_pipeline_are_deterministic_under_load runs either build).
#ifndef CAELO_XWG_FENCES
#define CAELO_XWG_FENCES 0
#endif
#if CAELO_XWG_FENCES
#define XWG_RELEASE() __threadfence()
#define XWG_ACQUIRE() __threadfence()
#define XWG_TICKET(PTR) __hip_atomic_fetch_add((PTR), 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
#else
#define XWG_RELEASE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define XWG_ACQUIRE() asm volatile("" ::: "memory")
#define XWG_TICKET(PTR) __hip_atomic_fetch_add((PTR), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif
__device__ inline void mm_publish(MmPartial *dst, const MmPartial &p) {
    unsigned long long *d = (unsigned long long *)dst;
    __hip_atomic_store(d + 0, (unsigned long long)__double_as_longlong(p.L1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d + 1, (unsigned long long)__double_as_longlong(p.L2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d + 2, (unsigned long long)__double_as_longlong(p.L3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d + 3, (unsigned long long)__double_as_longlong(p.U), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d + 4, ((unsigned long long)(unsigned)p.I2 << 32) | (unsigned)p.I1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline MmPartial mm_consume(const MmPartial *src) {
    unsigned long long *s = (unsigned long long *)src;
    MmPartial p;
    p.L1 = __longlong_as_double((long long)__hip_atomic_load(s + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    p.L2 = __longlong_as_double((long long)__hip_atomic_load(s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    p.L3 = __longlong_as_double((long long)__hip_atomic_load(s + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    p.U = __longlong_as_double((long long)__hip_atomic_load(s + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const unsigned long long ii = __hip_atomic_load(s + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    p.I1 = (int)(unsigned)(ii & 0xFFFFFFFFull);
    p.I2 = (int)(unsigned)(ii >> 32);
    return p;
}

CAELO_API int64_t caelo_match_ws_bytes(int64_t k1_max) {
    const int64_t tiles = (k1_max + 15) / 16;
    return 256 + ((tiles * 4 + 255) / 256) * 256 + tiles * MM_RS * 16 * (int64_t)sizeof(MmPartial);
}

__device__ inline double exact_dist(const float *a, const float *b, int dim) {
    double acc = 0.0;
    for (int c = 0; c < dim; ++c) {
        const double d = __dsub_rn((double)a[c], (double)b[c]);
        acc = __dadd_rn(acc, __dmul_rn(d, d));
    }
    return sqrt(acc);
}

// insert (lo, i) into an ascending top-3 (indices kept for the first two)
__device__ inline void top3_insert(double lo, int i, double &L1, double &L2, double &L3, int &I1, int &I2) {
    if (lo < L1) { L3 = L2; L2 = L1; I2 = I1; L1 = lo; I1 = i; }
    else if (lo < L2) { L3 = L2; L2 = lo; I2 = i; }
    else if (lo < L3) { L3 = lo; }
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

template <bool VEC>
__global__ void __launch_bounds__(64 * MM_WAVES) k_match_mfma(const caelo_pair_set ps, int ld0, int64_t k0_max, int ld1,
                                                              int64_t k1_max, int dim, size_t tbytes) {
    const caelo_pair_dev &P = ps.p[blockIdx.z];
    const float *__restrict__ f0 = P.f0, *__restrict__ f1 = P.f1;
    const int32_t *n0p = P.n0, *n1p = P.n1;
    int64_t *__restrict__ pair_idx = P.pair_idx;
    // workspace layout: stats [256 B] | tickets | partial results (see match_set)
    int32_t *stats = (int32_t *)P.ws_match;
    int32_t *tickets = (int32_t *)((char *)P.ws_match + 256);
    MmPartial *parts = (MmPartial *)((char *)P.ws_match + 256 + tbytes);
    __shared__ double sL[3][MM_WAVES][16];
    __shared__ int sI[2][MM_WAVES][16];
    __shared__ double sU[MM_WAVES][16];
    __shared__ int s_rescan[16];
    __shared__ double s_rd[MM_WAVES];
    __shared__ int s_ri[MM_WAVES];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, x = lane & 15;
    // counts live on the device; clamp so that a caller who forgot to order this launch after the
    // producer of n0/n1 reads garbage rows, never out of bounds
    const int k0 = n0p ? min(max(*n0p, 0), (int)k0_max) : (int)k0_max;
    const int k1 = n1p ? min(max(*n1p, 0), (int)k1_max) : (int)k1_max;
    const int ctile = blockIdx.x, rs = blockIdx.y;
    const int j0 = ctile * 16;
    if (j0 >= k1) return;  // uniform over the column tile's slices: no ticket needed
    if (k0 == 0) {         // no frame-0 descriptor at all (the reference's argmin would raise): index 0, the pose fails
        if (rs == 0 && tid < 16 && j0 + tid < k1) pair_idx[j0 + tid] = 0;
        return;
    }
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
    double L1 = BIG, L2 = BIG, L3 = BIG, U = BIG;
    int I1 = 0x7FFFFFFF, I2 = 0x7FFFFFFF;
    const int ntiles = (k0 + 15) >> 4;
    for (int t = rs * MM_WAVES + wave; t < ntiles; t += MM_RS * MM_WAVES) {
        const int i0 = t << 4;
        double a[MM_KSTEPS];
        load_frag<VEC>(f0 + (size_t)(i0 + x) * ld0, i0 + x < k0, g, dim, a);
        double p = 0.0;
#pragma unroll
        for (int s = 0; s < MM_KSTEPS; ++s) p += a[s] * a[s];
        p += __shfl_xor(p, 16);
        p += __shfl_xor(p, 32);  // |f0_{i0+x}|^2 on every lane with this x
        mm_f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < MM_KSTEPS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = g + 4 * r;          // f64 C/D layout: row = (lane >> 4) + 4 * reg, col = lane & 15
            const double n0 = __shfl(p, row);   // lane `row` (g = 0) holds that row's norm
            const int i = i0 + row;
            if (i < k0) {
                const double v = n0 - 2.0 * acc[r];
                const double e = kappa * (n0 + n1);
                U = (v + e) < U ? (v + e) : U;
                top3_insert(v - e, i, L1, L2, L3, I1, I2);
            }
        }
    }
    // ---- workgroup top-3 per column: merge the 4 lane groups by shuffles, the 16 waves through LDS
#define MM_SHFL_MERGE(OFF)                                                                           \
    {                                                                                                \
        const double pL1 = __shfl_xor(L1, OFF), pL2 = __shfl_xor(L2, OFF), pL3 = __shfl_xor(L3, OFF); \
        const double pU = __shfl_xor(U, OFF);                                                        \
        const int pI1 = __shfl_xor(I1, OFF), pI2 = __shfl_xor(I2, OFF);                              \
        U = pU < U ? pU : U;                                                                         \
        top3_insert(pL1, pI1, L1, L2, L3, I1, I2);                                                   \
        top3_insert(pL2, pI2, L1, L2, L3, I1, I2);                                                   \
        top3_insert(pL3, 0x7FFFFFFF, L1, L2, L3, I1, I2);                                            \
    }
    MM_SHFL_MERGE(16)
    MM_SHFL_MERGE(32)
    if (g == 0) {
        sL[0][wave][x] = L1; sL[1][wave][x] = L2; sL[2][wave][x] = L3;
        sI[0][wave][x] = I1; sI[1][wave][x] = I2;
        sU[wave][x] = U;
    }
    __syncthreads();
    MmPartial *mine = parts + ((size_t)ctile * MM_RS + rs) * 16;
    if (wave == 0) {
        // lane (g, x): merge waves 4g .. 4g+3 of column x, then the 4 lane groups again
        L1 = BIG; L2 = BIG; L3 = BIG; U = BIG; I1 = 0x7FFFFFFF; I2 = 0x7FFFFFFF;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int w = 4 * g + q;
            U = sU[w][x] < U ? sU[w][x] : U;
            top3_insert(sL[0][w][x], sI[0][w][x], L1, L2, L3, I1, I2);
            top3_insert(sL[1][w][x], sI[1][w][x], L1, L2, L3, I1, I2);
            top3_insert(sL[2][w][x], 0x7FFFFFFF, L1, L2, L3, I1, I2);
        }
        MM_SHFL_MERGE(16)
        MM_SHFL_MERGE(32)
        if (g == 0) {
            MmPartial pt;
            pt.L1 = L1; pt.L2 = L2; pt.L3 = L3; pt.U = U; pt.I1 = I1; pt.I2 = I2;
            mm_publish(&mine[x], pt);
        }
        XWG_RELEASE();  // the partial results have reached the coherence point before the ticket is taken
    }
    __syncthreads();
    if (tid == 0) {
        s_last = XWG_TICKET(&tickets[ctile]) == MM_RS - 1;
        if (s_last) tickets[ctile] = 0;  // self-cleaning: the workspace is ready for the next call, no memset launch
    }
    __syncthreads();
    if (!s_last) return;
    XWG_ACQUIRE();
    // ---- merge the slices and certify: one thread per column
    if (tid < 16) {
        const int j = j0 + tid;
        int rescan = 0;
        if (j < k1) {
            double a1 = BIG, a2 = BIG, a3 = BIG, Umin = BIG;
            int i1 = 0x7FFFFFFF, i2 = 0x7FFFFFFF;
            for (int q = 0; q < MM_RS; ++q) {
                const MmPartial pt = mm_consume(&parts[((size_t)ctile * MM_RS + q) * 16 + tid]);
                Umin = pt.U < Umin ? pt.U : Umin;
                top3_insert(pt.L1, pt.I1, a1, a2, a3, i1, i2);
                top3_insert(pt.L2, pt.I2, a1, a2, a3, i1, i2);
                top3_insert(pt.L3, 0x7FFFFFFF, a1, a2, a3, i1, i2);
            }
            if (a3 <= Umin) {
                rescan = 1;  // three or more rows inside the window
                atomicAdd(&stats[0], 1);
            } else if (a2 <= Umin) {
                atomicAdd(&stats[1], 1);
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
    for (int cidx = 0; cidx < 16; ++cidx) {
        if (!s_rescan[cidx]) continue;  // uniform
        const float *bj = f1 + (size_t)(j0 + cidx) * ld1;
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
            pair_idx[j0 + cidx] = besti;
        }
    }
}

CAELO_API int caelo_match(caelo_ctx *c, const float *f0, int ld0, int64_t k0_max, const int32_t *n0, const float *f1,
                          int ld1, int64_t k1_max, const int32_t *n1, int dim, int64_t *pair_idx, void *ws, void *stream) {
    CAELO_REQUIRE(c && f0 && f1 && pair_idx && ws, "null argument");
    caelo_pair_set ps = {};
    ps.n = 1;
    ps.p[0].f0 = f0; ps.p[0].n0 = n0; ps.p[0].f1 = f1; ps.p[0].n1 = n1; ps.p[0].pair_idx = pair_idx; ps.p[0].ws_match = ws;
    return match_set(ps, ld0, k0_max, ld1, k1_max, dim, caelo_stream(stream));
}

int match_set(const caelo_pair_set &ps, int ld0, int64_t k0_max, int ld1, int64_t k1_max, int dim, hipStream_t s) {
    CAELO_REQUIRE(ps.n >= 1 && ps.n <= CAELO_FB_MAX, "bad pair count");
    CAELO_REQUIRE(dim > 0 && dim <= MT_MAXDIM && ld0 >= dim && ld1 >= dim && k0_max > 0 && k1_max > 0, "bad shape");
    const int64_t tiles = (k1_max + 15) / 16;
    // layout: stats [256 B] | tickets | partial results.  The ticket region depends on k1_max: a workspace belongs to ONE
    // (stream, k1_max) -- a call with another k1_max would find the partial results of this one where its tickets live
    const size_t tbytes = (size_t)((tiles * 4 + 255) / 256) * 256;
    bool vec = (dim % 4 == 0) && (ld0 % 4 == 0) && (ld1 % 4 == 0);
    for (int i = 0; i < ps.n; ++i) vec = vec && (((uintptr_t)ps.p[i].f0 | (uintptr_t)ps.p[i].f1) & 15u) == 0;
    dim3 grid((unsigned)tiles, MM_RS, ps.n);
    if (vec)
        k_match_mfma<true><<<grid, 64 * MM_WAVES, 0, s>>>(ps, ld0, k0_max, ld1, k1_max, dim, tbytes);
    else
        k_match_mfma<false><<<grid, 64 * MM_WAVES, 0, s>>>(ps, ld0, k0_max, ld1, k1_max, dim, tbytes);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

// ------------------------------------------------------------------------------------------------
// 3x3 rigid fit from a cross-covariance H = sum (p1 - m1)(p0 - m0)^T   (Match.py:141-157)
// one-sided Jacobi SVD in f64: H V = U S ; R = V U^T (the reference's V.T @ U.T with V = Vh);
// det(R) < 0 -> the reference negates column 2 of Vh, i.e. R <- diag(1,1,-1) R  (:151-155).
// ------------------------------------------------------------------------------------------------
__device__ __noinline__ int rigid_from_H_jacobi(const double Hin[9], const double m0[3], const double m1[3], float R[9], float T[3]) {
    double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; ++i) A[i] = Hin[i];
    for (int sweep = 0; sweep < 12; ++sweep) {
        double offmax = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int r = 0; r < 3; ++r) {
                    alpha += A[3 * r + p] * A[3 * r + p];
                    beta += A[3 * r + q] * A[3 * r + q];
                    gamma += A[3 * r + p] * A[3 * r + q];
                }
                const double lim = 1e-30 + 1e-16 * sqrt(alpha * beta);
                if (fabs(gamma) <= lim) continue;
                offmax = fmax(offmax, fabs(gamma));
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int r = 0; r < 3; ++r) {
                    const double ap = A[3 * r + p], aq = A[3 * r + q];
                    A[3 * r + p] = cs * ap - sn * aq;
                    A[3 * r + q] = sn * ap + cs * aq;
                    const double vp = V[3 * r + p], vq = V[3 * r + q];
                    V[3 * r + p] = cs * vp - sn * vq;
                    V[3 * r + q] = sn * vp + cs * vq;
                }
            }
        if (offmax == 0.0) break;
    }
    // columns of A are u_i * s_i; order by descending s so a (near-)null direction ends up last
    double s[3];
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 3; ++i) s[i] = sqrt(A[i] * A[i] + A[3 + i] * A[3 + i] + A[6 + i] * A[6 + i]);
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (s[ord[b]] > s[ord[a]]) { const int tmp = ord[a]; ord[a] = ord[b]; ord[b] = tmp; }
    double U[9], W[9];
    for (int i = 0; i < 3; ++i) {
        const int cI = ord[i];
        const double inv = s[cI] > 0 ? 1.0 / s[cI] : 0.0;
        for (int r = 0; r < 3; ++r) { U[3 * r + i] = A[3 * r + cI] * inv; W[3 * r + i] = V[3 * r + cI]; }
    }
    const double tiny = 1e-12 * (s[ord[0]] > 0 ? s[ord[0]] : 1.0);
    if (s[ord[1]] <= tiny) {  // rank <= 1: any orthonormal completion (the pose is meaningless anyway)
        double e[3] = {1, 0, 0};
        if (fabs(U[0]) > 0.9) { e[0] = 0; e[1] = 1; }
        double d = e[0] * U[0] + e[1] * U[3] + e[2] * U[6];
        double v[3] = {e[0] - d * U[0], e[1] - d * U[3], e[2] - d * U[6]};
        const double nv = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        for (int r = 0; r < 3; ++r) U[3 * r + 1] = v[r] / nv;
    }
    if (s[ord[2]] <= tiny) {  // rank 2: u3 = u1 x u2 (sign is LAPACK-specific in the reference)
        U[2] = U[3] * U[7] - U[6] * U[4];
        U[5] = U[6] * U[1] - U[0] * U[7];
        U[8] = U[0] * U[4] - U[3] * U[1];
    }
    double Rd[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rd[3 * i + j] = W[3 * i] * U[3 * j] + W[3 * i + 1] * U[3 * j + 1] + W[3 * i + 2] * U[3 * j + 2];
    const double det = Rd[0] * (Rd[4] * Rd[8] - Rd[5] * Rd[7]) - Rd[1] * (Rd[3] * Rd[8] - Rd[5] * Rd[6]) +
                       Rd[2] * (Rd[3] * Rd[7] - Rd[4] * Rd[6]);
    if (det < 0) { Rd[6] = -Rd[6]; Rd[7] = -Rd[7]; Rd[8] = -Rd[8]; }  // :151-155
    for (int i = 0; i < 9; ++i) R[i] = (float)Rd[i];
    for (int i = 0; i < 3; ++i)
        T[i] = (float)(m0[i] - (Rd[3 * i] * m1[0] + Rd[3 * i + 1] * m1[1] + Rd[3 * i + 2] * m1[2]));  // :157
    return det < 0 ? -1 : 1;  // isCredible (:139,:152)
}

// Fast path: R = V U^T is the orthogonal polar factor of H^T.  Scaled Newton iteration
// X <- (g X + X^-T / g) / 2 (Higham) converges quadratically in f64 (5-7 steps, no sqrt/div chains of a
// Jacobi SVD: ~10x shorter dependency chain, and every hypothesis wavefront runs this serially).
// Rank-deficient or badly conditioned H (repeated sample indices) falls back to the Jacobi SVD above.
__device__ inline int rigid_from_H(const double H[9], const double m0[3], const double m1[3], float R[9], float T[3]) {
    double X[9] = {H[0], H[3], H[6], H[1], H[4], H[7], H[2], H[5], H[8]};  // X0 = H^T
    double fro = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) fro += X[i] * X[i];
    bool ok = fro > 0.0;
    double det0 = 0.0;
    for (int it = 0; it < 16 && ok; ++it) {
        double C[9];  // cofactors: X^-T = C / det
        C[0] = X[4] * X[8] - X[5] * X[7]; C[1] = X[5] * X[6] - X[3] * X[8]; C[2] = X[3] * X[7] - X[4] * X[6];
        C[3] = X[2] * X[7] - X[1] * X[8]; C[4] = X[0] * X[8] - X[2] * X[6]; C[5] = X[1] * X[6] - X[0] * X[7];
        C[6] = X[1] * X[5] - X[2] * X[4]; C[7] = X[2] * X[3] - X[0] * X[5]; C[8] = X[0] * X[4] - X[1] * X[3];
        const double det = X[0] * C[0] + X[1] * C[1] + X[2] * C[2];
        double nx = 0.0, nc = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) { nx += X[i] * X[i]; nc += C[i] * C[i]; }
        if (it == 0) {
            det0 = det;
            // sigma_min / sigma_max >= |det| / |X|_F^3 : refuse anything near rank deficiency
            if (!(fabs(det) > 1e-9 * nx * sqrt(nx))) { ok = false; break; }
        }
        const double inv = 1.0 / det;
        // gamma = sqrt(|X^-1|_F / |X|_F) only steers the convergence speed (any positive scaling has the same fixed
        // point, the polar factor): single precision is plenty and saves three f64 sqrt + one f64 divide per sweep
        const double g = (double)sqrtf(sqrtf((float)nc) * fabsf((float)inv) / sqrtf((float)nx));
        const double a = 0.5 * g, b = 0.5 * inv / g;
        double delta = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double nxt = a * X[i] + b * C[i];
            delta += (nxt - X[i]) * (nxt - X[i]);
            X[i] = nxt;
        }
        if (delta < 1e-30 * 3.0) break;  // |X_{k+1} - X_k|_F < 1e-15 |Q|_F
        if (it == 15) ok = false;
    }
    if (!ok) return rigid_from_H_jacobi(H, m0, m1, R, T);
    if (det0 < 0) { X[6] = -X[6]; X[7] = -X[7]; X[8] = -X[8]; }  // Match.py:151-155: Vh[:,2] *= -1  <=>  negate row 2 of R
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = (float)X[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        T[i] = (float)(m0[i] - (X[3 * i] * m1[0] + X[3 * i + 1] * m1[1] + X[3 * i + 2] * m1[2]));  // :157
    return det0 < 0 ? -1 : 1;
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
struct RansacWs {
    int32_t counts[CAELO_RANSAC_MAX_TRIALS];
    int32_t done;        // 1 once a level succeeded
    int32_t level_used;
    int32_t best_trial;  // within level_used
    int32_t iterations;
    int32_t success;
    float threshold;
    // The three below are zero between calls (the workspace is zero-filled once by its owner, every call leaves
    // it clean again -- no memset launch in the pair chain):
    int32_t arrived[CAELO_RANSAC_LEVELS];  // workgroups finished per level (the last one replays the rules, then resets it)
    int32_t finished;                      // the pose record is complete: later launches return at once
    int32_t exited;                        // last launch: workgroups that saw `finished`; the last of them resets both
};

CAELO_API int64_t caelo_ransac_ws_bytes(void) { return (int64_t)sizeof(RansacWs); }

// sequential accept / exit rules of Match.py:166-169,:181,:195-214 replayed over the counts (one wave:
// the counts are staged in LDS, lane 0 walks them)
__device__ void ransac_replay(int N, int level, RansacWs *ws, int *s_counts) {
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
        int v = i < CAELO_RANSAC_MAX_TRIALS ? __hip_atomic_load(&ws->counts[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
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
    const int success = (target > 0) || (least <= 0);  // leastInliers == 0 admits every hypothesis
    ws->iterations = stop;
    ws->threshold = 0.4f * (float)(1 << level);
    ws->success = success;
    ws->level_used = level;
    ws->best_trial = (success && target > 0) ? best : -1;  // N < 5: success without any accepted hypothesis -> identity (:177)
    if (success) ws->done = 1;
    (void)s_counts;
}

// hypothesis from 4 sampled pairs (SolveRT on the sample, Match.py:141-157; means / centring in f32 like
// np.mean on f32 rows, covariance in f64)
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
    rigid_from_H(H, m0, m1, R, T);
}

// One launch per threshold level (0.4 / 0.8 / 1.6 m).  Workgroup = 4 wavefronts = 4 hypotheses.
//   1. the matched pairs (P0[pair_idx[i]], P1[i]) are gathered once per workgroup into LDS (coalesced; the
//      per-hypothesis residual loops then never touch global memory);
//   2. one wavefront per hypothesis: Kabsch on the 4-sample, residuals, ballot + popcount inlier count;
//   3. the last workgroup to arrive (one ticket per workgroup; counts published with write-through
//      stores, read with agent-scope loads) replays the sequential accept rules in parallel;
//   4. if the level succeeded -- or it was the last one -- the same workgroup records the winner (recomputed from its
//      sample: R_star, T_star, threshold, iteration count).  Later level launches see `finished` and return at once.
// k_ransac_finish (one workgroup per pair, after the three level launches) writes the inlier mask of the winner and
// refits over all inliers (Match.py:273-282).  Round 1 did that inside the finishing workgroup of the level kernel,
// from the pairs it had staged in LDS; under concurrent streams ~1 call in 3 000 then stored 64-element runs of
// constant 0 / 1 from some of its wavefronts although the inlier COUNT summed from the same registers was right
// (tools/stress_mask.py: sentinel-filled buffer, D2H cross-check, exactly one replay per call counted) -- the
// "3 of 41 suite runs" of round 1.  The hand-off protocol was not involved.  A kernel of its own that reads the
// pairs from global memory shows 0 mismatches in 72 000 frames; DESIGN.md 4.4 has the measurements.
#define RE_WAVES 4
#define RE_LDS_PAIRS 1024

template <int level>
__global__ void __launch_bounds__(64 * RE_WAVES) k_ransac_level(const caelo_pair_set ps, int ld0, int ld1, int64_t k1_max) {
    const caelo_pair_dev &P = ps.p[blockIdx.z];
    const float *__restrict__ pc0 = P.pc0, *__restrict__ pc1 = P.pc1;
    const int64_t *__restrict__ pair_idx = P.pair_idx;
    const int32_t *n1p = P.n1;
    const double *__restrict__ rnd = P.rand;
    RansacWs *ws = (RansacWs *)P.ws_ransac;
    caelo_pose_result *res = P.result;
    __shared__ float sP0[RE_LDS_PAIRS * 3], sP1[RE_LDS_PAIRS * 3];
    __shared__ int s_counts[4];
    __shared__ int s_last, s_best, s_success;
    __shared__ float Rs[9], Ts[3];
    if (__hip_atomic_load(&ws->finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        // an earlier level completed the record.  In the last launch every workgroup passes through here, so the
        // last one to do so knows nobody will read `finished` again and clears it for the next call.
        if (level == CAELO_RANSAC_LEVELS - 1 && threadIdx.x == 0 && atomicAdd(&ws->exited, 1) == (int)gridDim.x - 1) {
            ws->exited = 0;
            __hip_atomic_store(&ws->finished, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const int N = n1p ? min(max(*n1p, 0), (int)k1_max) : (int)k1_max;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float thr = 0.4f * (float)(1 << level);  // 0.4, 0.8, 1.6 (:171,:210)
    // ---- 1. pairs -> LDS (falls back to the global arrays when they do not fit)
    const bool in_lds = N <= RE_LDS_PAIRS;
    if (in_lds) {
        for (int i = tid; i < N; i += 64 * RE_WAVES) {
            const float *a = pc0 + (size_t)ld0 * pair_idx[i];
            const float *b = pc1 + (size_t)ld1 * i;
            sP0[3 * i] = a[0]; sP0[3 * i + 1] = a[1]; sP0[3 * i + 2] = a[2];
            sP1[3 * i] = b[0]; sP1[3 * i + 1] = b[1]; sP1[3 * i + 2] = b[2];
        }
    }
    __syncthreads();
    const float *P0 = in_lds ? sP0 : pc0, *P1 = in_lds ? sP1 : pc1;
    const int l0 = in_lds ? 3 : ld0, l1 = in_lds ? 3 : ld1;
    const int64_t *pidx = in_lds ? nullptr : pair_idx;
    // ---- 2. this wavefront's hypothesis
    const int trial = blockIdx.x * RE_WAVES + wave;
    {
        float R[9], T[3];
        sample_hypothesis(P0, l0, pidx, P1, l1, N, rnd + ((size_t)level * CAELO_RANSAC_MAX_TRIALS + trial) * 4, R, T);
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
        if (lane == 0) {
            // read by the last workgroup of this launch: agent-scope store, complete before the arrival ticket; the
            // reader uses agent-scope loads (see XWG_* above)
            __hip_atomic_store(&ws->counts[trial], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            XWG_RELEASE();
        }
    }
    __syncthreads();
    if (tid == 0) {
        s_last = XWG_TICKET(&ws->arrived[level]) == CAELO_RANSAC_MAX_TRIALS / RE_WAVES - 1;
        if (s_last) ws->arrived[level] = 0;
    }
    __syncthreads();
    if (!s_last) return;
    XWG_ACQUIRE();
    // ---- 3. accept rules
    if (tid < 64) ransac_replay(N, level, ws, s_counts);
    __syncthreads();
    if (tid == 0) { s_success = ws->success; s_best = ws->best_trial; }
    __syncthreads();
    const int success = s_success, best = s_best;
    if (!success && level < CAELO_RANSAC_LEVELS - 1) return;  // escalate: the next launch doubles the threshold
    // ---- 4. finish: the winner is recomputed here (deterministic), identity if every level failed (:177)
    if (tid < 64) {
        float R[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}, T[3] = {0.f, 0.f, 0.f};
        if (best >= 0) sample_hypothesis(P0, l0, pidx, P1, l1, N, rnd + ((size_t)level * CAELO_RANSAC_MAX_TRIALS + best) * 4, R, T);
        if (tid < 9) Rs[tid] = R[tid];
        if (tid < 3) Ts[tid] = T[tid];
    }
    __syncthreads();
    if (tid < 9) res->R_ransac[tid] = Rs[tid];
    if (tid < 3) res->T_ransac[tid] = Ts[tid];
    if (tid == 0) {
        res->threshold = thr;
        res->success = success;
        res->iterations = ws->iterations;
        res->best_trial = best >= 0 ? level * CAELO_RANSAC_MAX_TRIALS + best : -1;
        res->n_pairs = N;
        if (level < CAELO_RANSAC_LEVELS - 1)  // the last level has no later launch to stop
            __hip_atomic_store(&ws->finished, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// After the three level launches: the inlier mask of the winning hypothesis (Match.py:193-194 with R_star, T_star and
// the final threshold), the inlier count and the refit over all inliers (Match.py:273-282).  One workgroup per pair.
__global__ void __launch_bounds__(256) k_ransac_finish(const caelo_pair_set ps, int ld0, int ld1, int64_t k1_max) {
    const caelo_pair_dev &P = ps.p[blockIdx.z];
    const float *__restrict__ pc0 = P.pc0, *__restrict__ pc1 = P.pc1;
    const int64_t *__restrict__ pair_idx = P.pair_idx;
    caelo_pose_result *res = P.result;
    uint8_t *mask = P.mask;
    __shared__ float Rs[9], Ts[3];
    __shared__ int s_nin;
    const int tid = threadIdx.x, lane = tid & 63;
    const int N = P.n1 ? min(max(*P.n1, 0), (int)k1_max) : (int)k1_max;
    if (tid < 9) Rs[tid] = res->R_ransac[tid];
    if (tid < 3) Ts[tid] = res->T_ransac[tid];
    if (tid == 0) s_nin = 0;
    __syncthreads();
    const float thr = res->threshold;
    const bool have = res->best_trial >= 0;
    int local = 0;
    // four consecutive pairs per thread, one aligned 32-bit store: the whole mask row goes out as full dwords
    const bool word_ok = (((uintptr_t)mask) & 3u) == 0;
    for (int i0 = tid * 4; i0 < (int)k1_max; i0 += 4 * 256) {
        unsigned int packed = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q;
            unsigned int in = 0;
            if (i < N && have) {
                const float *a = pc0 + (size_t)ld0 * pair_idx[i];
                const float *b = pc1 + (size_t)ld1 * i;
                in = residual(Rs, Ts, a[0], a[1], a[2], b[0], b[1], b[2]) < thr ? 1u : 0u;
            }
            packed |= in << (8 * q);
            local += (int)in;
        }
        if (word_ok && i0 + 3 < (int)k1_max) *(unsigned int *)(mask + i0) = packed;
        else
            for (int q = 0; q < 4 && i0 + q < (int)k1_max; ++q) mask[i0 + q] = (uint8_t)((packed >> (8 * q)) & 1u);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if (lane == 0) atomicAdd(&s_nin, local);
    __syncthreads();
    if (tid < 9) res->R[tid] = Rs[tid];
    if (tid < 3) res->T[tid] = Ts[tid];
    if (tid == 0) res->n_inliers = s_nin;
    __syncthreads();
    if (s_nin > 0) fit_block(pc0, ld0, pair_idx, pc1, ld1, mask, N, res->R, res->T, nullptr);  // :277-282
}

CAELO_API int caelo_ransac(caelo_ctx *c, const float *pc0, int ld0, const float *pc1, int ld1, const int64_t *pair_idx,
                           int64_t k1_max, const int32_t *n1, const double *rnd, caelo_pose_result *result,
                           uint8_t *inlier_mask, void *wsv, void *stream) {
    CAELO_REQUIRE(c && pc0 && pc1 && pair_idx && rnd && result && inlier_mask && wsv, "null argument");
    caelo_pair_set ps = {};
    ps.n = 1;
    caelo_pair_dev &p = ps.p[0];
    p.pc0 = pc0; p.pc1 = pc1; p.pair_idx = const_cast<int64_t *>(pair_idx); p.n1 = n1; p.rand = rnd; p.result = result;
    p.mask = inlier_mask; p.ws_ransac = wsv;
    return ransac_set(ps, ld0, ld1, k1_max, caelo_stream(stream));
}

int ransac_set(const caelo_pair_set &ps, int ld0, int ld1, int64_t k1_max, hipStream_t s) {
    CAELO_REQUIRE(ps.n >= 1 && ps.n <= CAELO_FB_MAX, "bad pair count");
    CAELO_REQUIRE(k1_max > 0 && ld0 >= 3 && ld1 >= 3, "bad shape");
    // the threshold level is a template parameter, not a kernel argument: the three launches then have IDENTICAL
    // argument blocks (see DESIGN.md 4.4 for the stale-argument observation that motivated this)
    const dim3 grid(CAELO_RANSAC_MAX_TRIALS / RE_WAVES, 1, ps.n);
    static_assert(CAELO_RANSAC_LEVELS == 3, "one instantiation per threshold level");
    k_ransac_level<0><<<grid, 64 * RE_WAVES, 0, s>>>(ps, ld0, ld1, k1_max);
    CAELO_LAUNCH_CHECK();
    k_ransac_level<1><<<grid, 64 * RE_WAVES, 0, s>>>(ps, ld0, ld1, k1_max);
    CAELO_LAUNCH_CHECK();
    k_ransac_level<2><<<grid, 64 * RE_WAVES, 0, s>>>(ps, ld0, ld1, k1_max);
    CAELO_LAUNCH_CHECK();
    k_ransac_finish<<<dim3(1, 1, ps.n), 256, 0, s>>>(ps, ld0, ld1, k1_max);
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
