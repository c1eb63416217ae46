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
// scored in parallel (one wavefront each, ballot + popcount for the inlier count) and a single
// thread replays the sequential accept / early-exit rules over the count array (verified equivalent
// on the reference: SURVEY 8a-9).
#include <stddef.h>

#include "caelo_internal.h"

// ------------------------------------------------------------------------------------------------
// NN match: exact f64 distances (the f32 GEMM form cannot guarantee the f64 argmin).
// Block = 4 waves, 16 frame-1 descriptors per block; F0 streamed through LDS in 64-row tiles.
// ------------------------------------------------------------------------------------------------
#define MT_J 4
#define MT_I 64
#define MT_MAXDIM 64

__global__ void __launch_bounds__(256) k_match(const float *__restrict__ f0, int ld0, int64_t k0_max, const int32_t *n0p,
                                               const float *__restrict__ f1, int ld1, int64_t k1_max, const int32_t *n1p,
                                               int dim, int64_t *__restrict__ pair_idx) {
    __shared__ float s0[MT_I * (MT_MAXDIM + 1)];
    __shared__ float s1[MT_J * MT_MAXDIM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = n0p ? *n0p : (int)k0_max;
    const int k1 = n1p ? *n1p : (int)k1_max;
    const int j0 = blockIdx.x * MT_J;
    if (j0 >= k1) return;
    for (int i = tid; i < MT_J * dim; i += 256) {
        const int j = i / dim, c = i % dim;
        s1[j * MT_MAXDIM + c] = (j0 + j < k1) ? f1[(size_t)(j0 + j) * ld1 + c] : 0.0f;
    }
    double best = 1.0e300;
    int besti = 0x7FFFFFFF;
    const int pitch = MT_MAXDIM + 1;
    for (int i0 = 0; i0 < k0; i0 += MT_I) {
        __syncthreads();
        for (int i = tid; i < MT_I * dim; i += 256) {
            const int r = i / dim, c = i % dim;
            s0[r * pitch + c] = (i0 + r < k0) ? f0[(size_t)(i0 + r) * ld0 + c] : 0.0f;
        }
        __syncthreads();
        if (i0 + lane < k0) {
            // this lane owns frame-0 row i0+lane; the wave owns frame-1 descriptor j0+wave
            double acc = 0.0;
            for (int c = 0; c < dim; ++c) {
                const double d = __dsub_rn((double)s0[lane * pitch + c], (double)s1[wave * MT_MAXDIM + c]);
                acc = __dadd_rn(acc, __dmul_rn(d, d));  // SciPy: s += d*d (no FMA)
            }
            const double dd = sqrt(acc);
            if (dd < best) { best = dd; besti = i0 + lane; }  // ascending i: first minimum kept
        }
    }
    // argmin across lanes (ties -> smaller index, i.e. the first minimum of np.argmin)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o);
        const int obi = __shfl_xor(besti, o);
        if (ob < best || (ob == best && obi < besti)) { best = ob; besti = obi; }
    }
    const int j = j0 + wave;
    if (lane == 0 && j < k1) pair_idx[j] = besti;
}

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
// Workgroup = 16 waves = one tile of 16 frame-1 descriptors; wave w scans frame-0 row tiles w, w+16, ...
// Each lane tracks the three smallest lower bounds of its stream; if a third one still passes the test
// the column is re-scanned exactly by the whole workgroup.
// ------------------------------------------------------------------------------------------------
typedef double mm_f64x4 __attribute__((ext_vector_type(4)));
#define MM_WAVES 16
#define MM_KSTEPS 16  // dim <= 64

__device__ inline double exact_dist(const float *a, const float *b, int dim) {
    double acc = 0.0;
    for (int c = 0; c < dim; ++c) {
        const double d = __dsub_rn((double)a[c], (double)b[c]);
        acc = __dadd_rn(acc, __dmul_rn(d, d));
    }
    return sqrt(acc);
}

__global__ void __launch_bounds__(64 * MM_WAVES) k_match_mfma(const float *__restrict__ f0, int ld0, int64_t k0_max,
                                                              const int32_t *n0p, const float *__restrict__ f1, int ld1,
                                                              int64_t k1_max, const int32_t *n1p, int dim,
                                                              int64_t *__restrict__ pair_idx) {
    __shared__ double sL[3][MM_WAVES * 4][16];  // three smallest lower bounds per (wave, g) stream and column
    __shared__ int sI[2][MM_WAVES * 4][16];     // rows of the two smallest
    __shared__ double sU[MM_WAVES * 4][16];     // smallest upper bound per stream
    __shared__ int s_rescan[16];
    __shared__ double s_rd[MM_WAVES];
    __shared__ int s_ri[MM_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, x = lane & 15;
    const int k0 = n0p ? *n0p : (int)k0_max;
    const int k1 = n1p ? *n1p : (int)k1_max;
    const int j0 = blockIdx.x * 16;
    if (j0 >= k1) return;
    const double kappa = (4.0 * (double)dim + 64.0) * 1.1102230246251565e-16;  // >= 2x the worst-case bound (dim + 20) 2^-53
    const double BIG = 1.0e300;
    // B fragments (this column tile) and |f1_j|^2
    double b[MM_KSTEPS];
    double n1 = 0.0;
#pragma unroll
    for (int s = 0; s < MM_KSTEPS; ++s) {
        const int c = 4 * s + g;
        b[s] = (j0 + x < k1 && c < dim) ? (double)f1[(size_t)(j0 + x) * ld1 + c] : 0.0;
        n1 += b[s] * b[s];
    }
    n1 += __shfl_xor(n1, 16);
    n1 += __shfl_xor(n1, 32);
    double L1 = BIG, L2 = BIG, L3 = BIG, U = BIG;
    int I1 = 0x7FFFFFFF, I2 = 0x7FFFFFFF;
    const int ntiles = (k0 + 15) >> 4;
    for (int t = wave; t < ntiles; t += MM_WAVES) {
        const int i0 = t << 4;
        double a[MM_KSTEPS];
        double p = 0.0;
#pragma unroll
        for (int s = 0; s < MM_KSTEPS; ++s) {
            const int c = 4 * s + g;
            a[s] = (i0 + x < k0 && c < dim) ? (double)f0[(size_t)(i0 + x) * ld0 + c] : 0.0;
            p += a[s] * a[s];
        }
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
                const double lo = v - e, up = v + e;
                U = up < U ? up : U;
                if (lo < L1) { L3 = L2; L2 = L1; I2 = I1; L1 = lo; I1 = i; }
                else if (lo < L2) { L3 = L2; L2 = lo; I2 = i; }
                else if (lo < L3) { L3 = lo; }
            }
        }
    }
    const int e = wave * 4 + g;
    sL[0][e][x] = L1; sL[1][e][x] = L2; sL[2][e][x] = L3;
    sI[0][e][x] = I1; sI[1][e][x] = I2;
    sU[e][x] = U;
    __syncthreads();
    // ---- one thread per column: certify
    if (tid < 16) {
        const int j = j0 + tid;
        int rescan = 0;
        if (j < k1) {
            double Umin = BIG;
            for (int q = 0; q < MM_WAVES * 4; ++q) Umin = sU[q][tid] < Umin ? sU[q][tid] : Umin;
            double best = BIG;
            int besti = 0x7FFFFFFF, ncand = 0, lasti = 0;
            const float *bj = f1 + (size_t)j * ld1;
            for (int q = 0; q < MM_WAVES * 4 && !rescan; ++q) {
                if (sL[2][q][tid] <= Umin) { rescan = 1; break; }
                for (int w = 0; w < 2; ++w) {
                    if (sL[w][q][tid] <= Umin) {
                        const int i = sI[w][q][tid];
                        if (ncand == 1) {  // a second survivor: evaluate the first one too
                            best = exact_dist(f0 + (size_t)lasti * ld0, bj, dim);
                            besti = lasti;
                        }
                        if (ncand >= 1) {
                            const double dd = exact_dist(f0 + (size_t)i * ld0, bj, dim);
                            if (dd < best || (dd == best && i < besti)) { best = dd; besti = i; }
                        }
                        ++ncand;
                        lasti = i;
                    }
                }
            }
            if (!rescan) pair_idx[j] = ncand == 1 ? lasti : besti;
        }
        s_rescan[tid] = rescan;
    }
    __syncthreads();
    // ---- exact re-scan of a column whose candidate list overflowed (whole workgroup)
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
                          int ld1, int64_t k1_max, const int32_t *n1, int dim, int64_t *pair_idx, void *stream) {
    CAELO_REQUIRE(c && f0 && f1 && pair_idx, "null argument");
    CAELO_REQUIRE(dim > 0 && dim <= MT_MAXDIM && ld0 >= dim && ld1 >= dim && k0_max > 0 && k1_max > 0, "bad shape");
    // (dim <= 64 always takes the MFMA path; the plain f64 kernel is kept for wider descriptors)
    if (dim <= 4 * MM_KSTEPS) {
        k_match_mfma<<<(unsigned)((k1_max + 15) / 16), 64 * MM_WAVES, 0, caelo_stream(stream)>>>(f0, ld0, k0_max, n0, f1, ld1,
                                                                                                   k1_max, n1, dim, pair_idx);
    } else {
        k_match<<<(unsigned)((k1_max + MT_J - 1) / MT_J), 256, 0, caelo_stream(stream)>>>(f0, ld0, k0_max, n0, f1, ld1,
                                                                                            k1_max, n1, dim, pair_idx);
    }
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}

// ------------------------------------------------------------------------------------------------
// 3x3 rigid fit from a cross-covariance H = sum (p1 - m1)(p0 - m0)^T   (Match.py:141-157)
// one-sided Jacobi SVD in f64: H V = U S ; R = V U^T (the reference's V.T @ U.T with V = Vh);
// det(R) < 0 -> the reference negates column 2 of Vh, i.e. R <- diag(1,1,-1) R  (:151-155).
// ------------------------------------------------------------------------------------------------
__device__ inline int rigid_from_H(const double Hin[9], const double m0[3], const double m1[3], float R[9], float T[3]) {
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
    float Rt[CAELO_RANSAC_MAX_TRIALS][12];
    int32_t done;        // 1 once a level succeeded (later levels early-exit)
    int32_t level_used;
    int32_t best_trial;  // within level_used
    int32_t iterations;
    int32_t success;
    float threshold;
    int32_t arrived[CAELO_RANSAC_LEVELS];  // hypotheses finished per level (last one replays the rules)
    int32_t pad;
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
        int v = i < CAELO_RANSAC_MAX_TRIALS ? ws->counts[i] : 0;
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
    ws->best_trial = success ? best : -1;
    if (success) ws->done = 1;
    (void)s_counts;
}

// one wavefront per hypothesis; the last wavefront of a level to finish replays the accept rules
__global__ void __launch_bounds__(64) k_ransac_eval(const float *__restrict__ pc0, int ld0, const float *__restrict__ pc1,
                                                    int ld1, const int64_t *__restrict__ pair_idx, int64_t k1_max,
                                                    const int32_t *n1p, const double *__restrict__ rnd, int level,
                                                    RansacWs *ws) {
    __shared__ int s_counts[CAELO_RANSAC_MAX_TRIALS];
    __shared__ int s_last;
    if (__hip_atomic_load(&ws->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const int N = n1p ? *n1p : (int)k1_max;
    const int trial = blockIdx.x;
    const int lane = threadIdx.x;
    const float thr = 0.4f * (float)(1 << level);  // 0.4, 0.8, 1.6 (:171,:210)
    const double *r4 = rnd + ((size_t)level * CAELO_RANSAC_MAX_TRIALS + trial) * 4;
    // ---- 4-point sample with replacement (:182-184): idx = int32(u * N)
    float s0[4][3], s1[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = (int)(r4[q] * (double)N);
        const int64_t i0 = pair_idx[idx];
#pragma unroll
        for (int a = 0; a < 3; ++a) { s0[q][a] = pc0[(size_t)ld0 * i0 + a]; s1[q][a] = pc1[(size_t)ld1 * idx + a]; }
    }
    // SolveRT on the sample (:141-157).  means/centering in f32 like np.mean on f32 rows.
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
    float R[9], T[3];
    rigid_from_H(H, m0, m1, R, T);
    // ---- residuals + inlier count (:191-194): ballot + popcount per 64 pairs
    int cnt = 0;
    for (int i = lane; i < ((N + 63) & ~63); i += 64) {
        bool in = false;
        if (i < N) {
            const float *a = pc0 + (size_t)ld0 * pair_idx[i];
            const float *b = pc1 + (size_t)ld1 * i;
            in = residual(R, T, a[0], a[1], a[2], b[0], b[1], b[2]) < thr;
        }
        cnt += __popcll(__ballot(in));
    }
    if (lane == 0) {
        ws->counts[trial] = cnt;
#pragma unroll
        for (int i = 0; i < 9; ++i) ws->Rt[trial][i] = R[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) ws->Rt[trial][9 + i] = T[i];
        __threadfence();  // release the count before the arrival ticket
        s_last = atomicAdd(&ws->arrived[level], 1) == CAELO_RANSAC_MAX_TRIALS - 1;
    }
    __syncthreads();
    if (s_last) {
        __threadfence();  // acquire: every other hypothesis' count is visible
        ransac_replay(N, level, ws, s_counts);
    }
}

// inlier mask of the winner, then the refit over all inliers (Match.py:273-282)
__global__ void __launch_bounds__(256) k_ransac_finish(const float *__restrict__ pc0, int ld0, const float *__restrict__ pc1,
                                                       int ld1, const int64_t *__restrict__ pair_idx, int64_t k1_max,
                                                       const int32_t *n1p, RansacWs *ws, caelo_pose_result *res,
                                                       uint8_t *mask) {
    __shared__ float Rs[9], Ts[3];
    __shared__ int n_in;
    const int N = n1p ? *n1p : (int)k1_max;
    const int tid = threadIdx.x;
    const int best = ws->best_trial;
    const float thr = ws->threshold;
    if (tid < 9) Rs[tid] = best >= 0 ? ws->Rt[best][tid] : ((tid % 4 == 0) ? 1.0f : 0.0f);  // :177 identity
    if (tid < 3) Ts[tid] = best >= 0 ? ws->Rt[best][9 + tid] : 0.0f;
    if (tid == 0) n_in = 0;
    __syncthreads();
    int local = 0;
    for (int i = tid; i < (int)k1_max; i += 256) {
        uint8_t in = 0;
        if (i < N && best >= 0) {
            const float *a = pc0 + (size_t)ld0 * pair_idx[i];
            const float *b = pc1 + (size_t)ld1 * i;
            in = residual(Rs, Ts, a[0], a[1], a[2], b[0], b[1], b[2]) < thr;
        }
        mask[i] = in;
        local += in;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((tid & 63) == 0) atomicAdd(&n_in, local);
    __syncthreads();
    if (tid < 9) { res->R_ransac[tid] = Rs[tid]; res->R[tid] = Rs[tid]; }
    if (tid < 3) { res->T_ransac[tid] = Ts[tid]; res->T[tid] = Ts[tid]; }
    if (tid == 0) {
        res->threshold = thr;
        res->success = ws->success;
        res->iterations = ws->iterations;
        res->n_inliers = n_in;
        res->best_trial = best >= 0 ? ws->level_used * CAELO_RANSAC_MAX_TRIALS + best : -1;
        res->n_pairs = N;
    }
    __syncthreads();
    if (n_in > 0) fit_block(pc0, ld0, pair_idx, pc1, ld1, mask, N, res->R, res->T, nullptr);  // :277-282
}

CAELO_API int caelo_ransac(caelo_ctx *c, const float *pc0, int ld0, const float *pc1, int ld1, const int64_t *pair_idx,
                           int64_t k1_max, const int32_t *n1, const double *rnd, caelo_pose_result *result,
                           uint8_t *inlier_mask, void *wsv, void *stream) {
    CAELO_REQUIRE(c && pc0 && pc1 && pair_idx && rnd && result && inlier_mask && wsv, "null argument");
    CAELO_REQUIRE(k1_max > 0 && ld0 >= 3 && ld1 >= 3, "bad shape");
    hipStream_t s = caelo_stream(stream);
    RansacWs *ws = (RansacWs *)wsv;
    CAELO_HIP(hipMemsetAsync(&ws->done, 0, sizeof(RansacWs) - offsetof(RansacWs, done), s));
    for (int level = 0; level < CAELO_RANSAC_LEVELS; ++level) {
        k_ransac_eval<<<CAELO_RANSAC_MAX_TRIALS, 64, 0, s>>>(pc0, ld0, pc1, ld1, pair_idx, k1_max, n1, rnd, level, ws);
        CAELO_LAUNCH_CHECK();
    }
    k_ransac_finish<<<1, 256, 0, s>>>(pc0, ld0, pc1, ld1, pair_idx, k1_max, n1, ws, result, inlier_mask);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}
