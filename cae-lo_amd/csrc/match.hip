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
#include "caelo_internal.h"

// ------------------------------------------------------------------------------------------------
// NN match: exact f64 distances (the f32 GEMM form cannot guarantee the f64 argmin).
// Block = 4 waves, 16 frame-1 descriptors per block; F0 streamed through LDS in 64-row tiles.
// ------------------------------------------------------------------------------------------------
#define MT_J 16
#define MT_I 64
#define MT_MAXDIM 64

__global__ void __launch_bounds__(256) k_match(const float *__restrict__ f0, int64_t k0_max, const int32_t *n0p,
                                               const float *__restrict__ f1, int64_t k1_max, const int32_t *n1p, int dim,
                                               int64_t *__restrict__ pair_idx) {
    __shared__ float s0[MT_I * (MT_MAXDIM + 1)];
    __shared__ float s1[MT_J * MT_MAXDIM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = n0p ? *n0p : (int)k0_max;
    const int k1 = n1p ? *n1p : (int)k1_max;
    const int j0 = blockIdx.x * MT_J;
    if (j0 >= k1) return;
    for (int i = tid; i < MT_J * dim; i += 256) {
        const int j = i / dim, c = i % dim;
        s1[j * MT_MAXDIM + c] = (j0 + j < k1) ? f1[(size_t)(j0 + j) * dim + c] : 0.0f;
    }
    double best[4];
    int besti[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { best[q] = 1.0e300; besti[q] = 0x7FFFFFFF; }
    const int pitch = MT_MAXDIM + 1;
    for (int i0 = 0; i0 < k0; i0 += MT_I) {
        __syncthreads();
        for (int i = tid; i < MT_I * dim; i += 256) {
            const int r = i / dim, c = i % dim;
            s0[r * pitch + c] = (i0 + r < k0) ? f0[(size_t)(i0 + r) * dim + c] : 0.0f;
        }
        __syncthreads();
        if (i0 + lane < k0) {
            // this lane owns frame-0 row i0+lane; the wave's four frame-1 descriptors are 4*wave..+3
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            for (int c = 0; c < dim; ++c) {
                const double a = (double)s0[lane * pitch + c];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double d = __dsub_rn(a, (double)s1[(wave * 4 + q) * MT_MAXDIM + c]);
                    acc[q] = __dadd_rn(acc[q], __dmul_rn(d, d));  // SciPy: s += d*d (no FMA)
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double dd = sqrt(acc[q]);
                if (dd < best[q]) { best[q] = dd; besti[q] = i0 + lane; }  // ascending i: first minimum kept
            }
        }
    }
    // argmin across lanes (ties -> smaller index, i.e. the first minimum of np.argmin)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double b = best[q];
        int bi = besti[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(b, o);
            const int obi = __shfl_xor(bi, o);
            if (ob < b || (ob == b && obi < bi)) { b = ob; bi = obi; }
        }
        const int j = j0 + wave * 4 + q;
        if (lane == 0 && j < k1) pair_idx[j] = bi;
    }
}

CAELO_API int caelo_match(caelo_ctx *c, const float *f0, int64_t k0_max, const int32_t *n0, const float *f1,
                          int64_t k1_max, const int32_t *n1, int dim, int64_t *pair_idx, void *stream) {
    CAELO_REQUIRE(c && f0 && f1 && pair_idx, "null argument");
    CAELO_REQUIRE(dim > 0 && dim <= MT_MAXDIM && k0_max > 0 && k1_max > 0, "bad shape");
    k_match<<<(unsigned)((k1_max + MT_J - 1) / MT_J), 256, 0, caelo_stream(stream)>>>(f0, k0_max, n0, f1, k1_max, n1, dim,
                                                                                        pair_idx);
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
// generic SolveRT over n pairs (one workgroup)
// ------------------------------------------------------------------------------------------------
__device__ inline double block_sum(double v, double *scratch) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    double s = 0;
    for (unsigned w = 0; w < blockDim.x / 64; ++w) s += scratch[w];
    return s;
}

// gather-aware fit: pair i is (p0[idx0 ? idx0[i] : i], p1[idx1 ? idx1[i] : i]) restricted to mask
__device__ void fit_block(const float *p0, const int64_t *idx0, const float *p1, const uint8_t *mask, int n, float *R,
                          float *T, int *credible) {
    __shared__ double scratch[16];
    const int tid = threadIdx.x;
    double c = 0, a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
    for (int i = tid; i < n; i += blockDim.x) {
        if (mask && !mask[i]) continue;
        const float *u = p0 + 3 * (idx0 ? idx0[i] : i);
        const float *v = p1 + 3 * (int64_t)i;
        c += 1; a0 += u[0]; a1 += u[1]; a2 += u[2]; b0 += v[0]; b1 += v[1]; b2 += v[2];
    }
    c = block_sum(c, scratch);
    double m0[3], m1[3];
    m0[0] = block_sum(a0, scratch); m0[1] = block_sum(a1, scratch); m0[2] = block_sum(a2, scratch);
    m1[0] = block_sum(b0, scratch); m1[1] = block_sum(b1, scratch); m1[2] = block_sum(b2, scratch);
    if (c < 1) return;
    for (int i = 0; i < 3; ++i) { m0[i] /= c; m1[i] /= c; }
    double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < n; i += blockDim.x) {
        if (mask && !mask[i]) continue;
        const float *u = p0 + 3 * (idx0 ? idx0[i] : i);
        const float *v = p1 + 3 * (int64_t)i;
        const double x1 = v[0] - m1[0], y1 = v[1] - m1[1], z1 = v[2] - m1[2];
        const double x0 = u[0] - m0[0], y0 = u[1] - m0[1], z0 = u[2] - m0[2];
        h[0] += x1 * x0; h[1] += x1 * y0; h[2] += x1 * z0;  // H = P1c^T P0c  (:146)
        h[3] += y1 * x0; h[4] += y1 * y0; h[5] += y1 * z0;
        h[6] += z1 * x0; h[7] += z1 * y0; h[8] += z1 * z0;
    }
    double H[9];
    for (int i = 0; i < 9; ++i) H[i] = block_sum(h[i], scratch);
    if (tid == 0) {
        const int cred = rigid_from_H(H, m0, m1, R, T);
        if (credible) *credible = cred;
    }
}

__global__ void __launch_bounds__(256) k_solve_rt(const float *p0, const float *p1, int n, float *R, float *T, int *credible) {
    fit_block(p0, nullptr, p1, nullptr, n, R, T, credible);
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
};

CAELO_API int64_t caelo_ransac_ws_bytes(void) { return (int64_t)sizeof(RansacWs); }

// one wavefront per hypothesis
__global__ void __launch_bounds__(64) k_ransac_eval(const float *__restrict__ pc0, const float *__restrict__ pc1,
                                                    const int64_t *__restrict__ pair_idx, int64_t k1_max,
                                                    const int32_t *n1p, const double *__restrict__ rnd, int level,
                                                    RansacWs *ws) {
    if (ws->done) return;
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
        for (int a = 0; a < 3; ++a) { s0[q][a] = pc0[3 * i0 + a]; s1[q][a] = pc1[3 * (int64_t)idx + a]; }
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
            const int64_t i0 = pair_idx[i];
            in = residual(R, T, pc0[3 * i0], pc0[3 * i0 + 1], pc0[3 * i0 + 2], pc1[3 * (int64_t)i], pc1[3 * (int64_t)i + 1],
                          pc1[3 * (int64_t)i + 2]) < thr;
        }
        cnt += __popcll(__ballot(in));
    }
    if (lane == 0) {
        ws->counts[trial] = cnt;
#pragma unroll
        for (int i = 0; i < 9; ++i) ws->Rt[trial][i] = R[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) ws->Rt[trial][9 + i] = T[i];
    }
}

// sequential accept / exit rules of Match.py:166-169,:181,:195-214 replayed over the counts
__global__ void k_ransac_replay(int64_t k1_max, const int32_t *n1p, int level, RansacWs *ws) {
    if (ws->done) return;
    const int N = n1p ? *n1p : (int)k1_max;
    const int least = (100 < (int)(0.2 * N)) ? 100 : (int)(0.2 * N);  // :166
    const double min_success = 0.25 * N;                              // :167
    int it = 0, cur = 0, best = -1, success = 0;
    while (it < 100 || (it < CAELO_RANSAC_MAX_TRIALS && (double)cur < min_success)) {  // :181
        const int c = ws->counts[it];
        if (c >= least) {
            if (c > cur) { cur = c; best = it; }  // strict > (:199)
            success = 1;
        }
        ++it;
    }
    ws->iterations = it;
    ws->threshold = 0.4f * (float)(1 << level);
    if (success) {
        ws->done = 1;
        ws->success = 1;
        ws->level_used = level;
        ws->best_trial = best;
    } else {
        ws->success = 0;
        ws->level_used = level;
        ws->best_trial = -1;
    }
}

// inlier mask of the winner, then the refit over all inliers (Match.py:273-282)
__global__ void __launch_bounds__(256) k_ransac_finish(const float *__restrict__ pc0, const float *__restrict__ pc1,
                                                       const int64_t *__restrict__ pair_idx, int64_t k1_max,
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
            const int64_t i0 = pair_idx[i];
            in = residual(Rs, Ts, pc0[3 * i0], pc0[3 * i0 + 1], pc0[3 * i0 + 2], pc1[3 * (int64_t)i], pc1[3 * (int64_t)i + 1],
                          pc1[3 * (int64_t)i + 2]) < thr;
        }
        mask[i] = in;
        local += in;
    }
    atomicAdd(&n_in, local);
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
    if (n_in > 0) fit_block(pc0, pair_idx, pc1, mask, N, res->R, res->T, nullptr);  // :277-282
}

CAELO_API int caelo_ransac(caelo_ctx *c, const float *pc0, const float *pc1, const int64_t *pair_idx, int64_t k1_max,
                           const int32_t *n1, const double *rnd, caelo_pose_result *result, uint8_t *inlier_mask, void *wsv,
                           void *stream) {
    CAELO_REQUIRE(c && pc0 && pc1 && pair_idx && rnd && result && inlier_mask && wsv, "null argument");
    CAELO_REQUIRE(k1_max > 0, "bad shape");
    hipStream_t s = caelo_stream(stream);
    RansacWs *ws = (RansacWs *)wsv;
    CAELO_HIP(hipMemsetAsync(&ws->done, 0, sizeof(int32_t) * 5 + sizeof(float), s));
    for (int level = 0; level < CAELO_RANSAC_LEVELS; ++level) {
        k_ransac_eval<<<CAELO_RANSAC_MAX_TRIALS, 64, 0, s>>>(pc0, pc1, pair_idx, k1_max, n1, rnd, level, ws);
        CAELO_LAUNCH_CHECK();
        k_ransac_replay<<<1, 1, 0, s>>>(k1_max, n1, level, ws);
        CAELO_LAUNCH_CHECK();
    }
    k_ransac_finish<<<1, 256, 0, s>>>(pc0, pc1, pair_idx, k1_max, n1, ws, result, inlier_mask);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}
