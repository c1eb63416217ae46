// certify.hip -- the HOST half of the bit-exact RANSAC (SURVEY 8a-8 .. a-10; Match.py:138-218, :269-283).  No kernels here.
//
// Why a host half exists.  The reference fits every 4-point hypothesis with NumPy: float32 means, centring and covariance
// (np.dot -> cblas_sgemm), np.linalg.svd (which computes in FLOAT64 whatever the input type: dgesdd, results cast to
// float32), R = np.dot(V.T, U.T) (sgemm), T through sgemv, residuals through sgemm -- Match.py:141-157,:190-193.  The bits of
// R and T therefore depend on the BLAS that NumPy was built with (accumulation order, fused or unfused multiply-adds: an
// OpenBLAS Haswell kernel and a reference BLAS differ in the last bit of a third of the residuals), and a residual within
// ~1e-5 m of the threshold falls on either side of it.  The kernels of match.hip fit in float64 and score all 500
// hypotheses; beside each count they give a rigorous UPPER BOUND on the count the reference's arithmetic can reach
// (k_ransac_hyp, `hi`).  This file replays the sequential accept / exit rules of Match.py:181-214 over those bounds and
// re-evaluates, through the very cblas_sgemm / cblas_sgemv / dgesdd entry points the process's NumPy calls (handed over by
// caelo/hostblas.py: caelo_host_bind_blas), only the hypotheses that can decide -- the running winner, one to three per pair --
// until the winner is exact and no other hypothesis' bound reaches it.  The inlier mask, R_star / T_star and the refit over
// the inliers (Match.py:273-282) then come from the same calls: bit for bit what the reference computes on this host.
//
// Nothing here is a fallback for the device path: a pair costs ~10 us of one host core against the 500 hypotheses x 1024
// residuals the GPU has scored, and without the device's bounds caelo_host_ransac evaluates every hypothesis (used by the
// API for more than 1024 pairs and by the tests as the cross-check of the bounds).
#include "caelo_internal.h"

#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ---- the BLAS / LAPACK of the process's NumPy ------------------------------------------------------------------------
// cblas enums (cblas.h)
enum { kRowMajor = 101, kNoTrans = 111, kTrans = 112 };
typedef void (*sgemm32_t)(int, int, int, int, int, int, float, const float *, int, const float *, int, float, float *, int);
typedef void (*sgemm64_t)(int, int, int, int64_t, int64_t, int64_t, float, const float *, int64_t, const float *, int64_t, float,
                          float *, int64_t);
typedef void (*sgemv32_t)(int, int, int, int, float, const float *, int, const float *, int, float, float *, int);
typedef void (*sgemv64_t)(int, int, int64_t, int64_t, float, const float *, int64_t, const float *, int64_t, float, float *, int64_t);
// Fortran dgesdd (hidden string length last)
typedef void (*dgesdd32_t)(const char *, const int *, const int *, double *, const int *, double *, double *, const int *, double *,
                           const int *, double *, const int *, int *, int *, size_t);
typedef void (*dgesdd64_t)(const char *, const int64_t *, const int64_t *, double *, const int64_t *, double *, double *,
                           const int64_t *, double *, const int64_t *, double *, const int64_t *, int64_t *, int64_t *, size_t);

struct HostBlas {
    void *sgemm = nullptr, *sgemv = nullptr, *dgesdd = nullptr;
    int ilp64 = 0;
    int lwork = 0;  // dgesdd's own answer to the workspace query for a 3 x 3 matrix (what NumPy allocates)
};
HostBlas g_blas;
std::atomic<int64_t> g_bound_violations{0};   // exact counts above the device's bound (ransac_host)

inline void gemm(int ta, int tb, int64_t m, int64_t n, int64_t k, const float *a, int64_t lda, const float *b, int64_t ldb, float *c,
                 int64_t ldc) {
    if (g_blas.ilp64)
        ((sgemm64_t)g_blas.sgemm)(kRowMajor, ta, tb, m, n, k, 1.0f, a, lda, b, ldb, 0.0f, c, ldc);
    else
        ((sgemm32_t)g_blas.sgemm)(kRowMajor, ta, tb, (int)m, (int)n, (int)k, 1.0f, a, (int)lda, b, (int)ldb, 0.0f, c, (int)ldc);
}
inline void gemv3(const float *a, const float *x, float *y) {  // y = A x, A [3][3] row-major (NumPy: matrix times column)
    if (g_blas.ilp64)
        ((sgemv64_t)g_blas.sgemv)(kRowMajor, kNoTrans, 3, 3, 1.0f, a, 3, x, 1, 0.0f, y, 1);
    else
        ((sgemv32_t)g_blas.sgemv)(kRowMajor, kNoTrans, 3, 3, 1.0f, a, 3, x, 1, 0.0f, y, 1);
}
// np.linalg.svd(H) for a float32 [3][3] H: float64 copy in Fortran order, dgesdd('A'), U / Vh cast to float32 (C order)
inline bool svd3(const float *H, float *U, float *Vh, int lwork_query_only = 0) {
    double a[9], s[3], u[9], vt[9], work[512];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) a[i + 3 * j] = (double)H[3 * i + j];
    const char jobz = 'A';
    if (g_blas.ilp64) {
        const int64_t n3 = 3;
        int64_t lw = lwork_query_only ? -1 : g_blas.lwork, info = 0, iwork[24];
        ((dgesdd64_t)g_blas.dgesdd)(&jobz, &n3, &n3, a, &n3, s, u, &n3, vt, &n3, work, &lw, iwork, &info, 1);
        if (info != 0) return false;
    } else {
        const int n3 = 3;
        int lw = lwork_query_only ? -1 : g_blas.lwork, info = 0, iwork[24];
        ((dgesdd32_t)g_blas.dgesdd)(&jobz, &n3, &n3, a, &n3, s, u, &n3, vt, &n3, work, &lw, iwork, &info, 1);
        if (info != 0) return false;
    }
    if (lwork_query_only) {
        g_blas.lwork = (int)work[0];
        return g_blas.lwork >= 1 && g_blas.lwork <= 512;
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            U[3 * i + j] = (float)u[i + 3 * j];
            Vh[3 * i + j] = (float)vt[i + 3 * j];
        }
    return true;
}

// ---- SolveRT, statement by statement (Match.py:138-158) ---------------------------------------------------------------
// rows(i): the i-th pair's points.  scratch c0 / c1: [n][3] floats each.
struct Fit {
    float R[9], T[3];
    int credible;
};
bool solve_rt_rows(const float *p0, const float *p1, int64_t n, float *c0, float *c1, Fit *out) {
    // np.mean(axis=0) of a C-contiguous [n][3] float32: the rows are added one after the other in float32; the division is
    // NumPy 2's (float32 array / intp scalar -> float64 loop, result cast back): exact for n = 4 either way
    float m0[3], m1[3];
    for (int a = 0; a < 3; ++a) {
        float s0 = p0[a], s1 = p1[a];
        for (int64_t i = 1; i < n; ++i) { s0 = s0 + p0[3 * i + a]; s1 = s1 + p1[3 * i + a]; }
        m0[a] = (float)((double)s0 / (double)n);
        m1[a] = (float)((double)s1 / (double)n);
    }
    for (int64_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) { c0[3 * i + a] = p0[3 * i + a] - m0[a]; c1[3 * i + a] = p1[3 * i + a] - m1[a]; }
    float H[9], U[9], V[9];
    gemm(kTrans, kNoTrans, 3, 3, n, c1, 3, c0, 3, H, 3);  // :146 np.dot(P1.T, P0)
    if (!svd3(H, U, V)) return false;                     // :148
    gemm(kTrans, kTrans, 3, 3, 3, V, 3, U, 3, out->R, 3);  // :149 np.dot(V.T, U.T)
    const float *R = out->R;
    const double det = (double)R[0] * ((double)R[4] * R[8] - (double)R[5] * R[7]) - (double)R[1] * ((double)R[3] * R[8] - (double)R[5] * R[6]) +
                       (double)R[2] * ((double)R[3] * R[7] - (double)R[4] * R[6]);
    out->credible = 1;
    if (det < 0) {  // :151-155: V[:, 2] *= -1 -- a COLUMN of Vh
        out->credible = -1;
        V[2] = -V[2]; V[5] = -V[5]; V[8] = -V[8];
        gemm(kTrans, kTrans, 3, 3, 3, V, 3, U, 3, out->R, 3);
    }
    float y[3];
    gemv3(out->R, m1, y);  // :157 np.dot(R, mean1.T)
    for (int a = 0; a < 3; ++a) out->T[a] = m0[a] - y[a];
    return true;
}

// residuals and inlier mask of one pose (Match.py:190-193); x: scratch [3][n] floats.  Returns the inlier count.
int score(const Fit &f, const float *P0, const float *P1, int64_t n, float thr, float *x, uint8_t *mask) {
    gemm(kNoTrans, kTrans, 3, n, 3, f.R, 3, P1, 3, x, n);  // np.dot(R, Pairs1.T)
    int cnt = 0;
    for (int64_t i = 0; i < n; ++i) {
        const float dx = P0[3 * i] - (x[i] + f.T[0]), dy = P0[3 * i + 1] - (x[n + i] + f.T[1]), dz = P0[3 * i + 2] - (x[2 * n + i] + f.T[2]);
        const float d = sqrtf((dx * dx + dy * dy) + dz * dz);  // np.linalg.norm(axis=1): sqrt(add.reduce(x * x)), three terms in order
        const bool in = d < thr;
        if (mask) mask[i] = in ? 1 : 0;
        cnt += in ? 1 : 0;
    }
    return cnt;
}

struct Scratch {
    std::vector<float> c0, c1, x, s0, s1;
    std::vector<uint8_t> m_try, m_best;
    std::vector<int32_t> cnt;
    std::vector<uint8_t> exact;
    void size(int64_t n) {
        if ((int64_t)x.size() < 3 * n) { c0.resize(3 * n); c1.resize(3 * n); x.resize(3 * n); s0.resize(3 * n); s1.resize(3 * n); }
        if ((int64_t)m_try.size() < n) { m_try.resize(n); m_best.resize(n); }
        cnt.resize(CAELO_RANSAC_MAX_TRIALS);
        exact.resize(CAELO_RANSAC_MAX_TRIALS);
    }
};

// one hypothesis of Match.py:182-194 from its four indices
bool hypothesis(const float *P0, const float *P1, int64_t n, const int32_t idx[4], float thr, Scratch &S, Fit *fit, uint8_t *mask, int *count) {
    float s0[12], s1[12], c0[12], c1[12];
    for (int q = 0; q < 4; ++q) {
        const int64_t i = idx[q];
        if (i < 0 || i >= n) return false;
        for (int a = 0; a < 3; ++a) { s0[3 * q + a] = P0[3 * i + a]; s1[3 * q + a] = P1[3 * i + a]; }
    }
    if (!solve_rt_rows(s0, s1, 4, c0, c1, fit)) return false;
    *count = score(*fit, P0, P1, n, thr, S.x.data(), mask);
    return true;
}

// the sequential rules of Match.py:181-206 over counts (exact or upper bounds): -> winner (-1: none), iterations
void replay(const int32_t *c, int64_t n_pairs, int *winner, int *iters) {
    const int least = (100 < (int)(0.2 * (double)n_pairs)) ? 100 : (int)(0.2 * (double)n_pairs);  // :166
    const double min_success = 0.25 * (double)n_pairs;                                               // :167
    int best = 0, w = -1, it = 0;
    while (it < 100 || (it < CAELO_RANSAC_MAX_TRIALS && (double)best < min_success)) {  // :181 (minTrails 100, maxTrails 500)
        const int v = c[it];
        if (v >= least && v > best) { best = v; w = it; }
        ++it;
    }
    *winner = w;
    *iters = it;
}

struct Outcome {
    int success = 0, iterations = 0, best_trial = -1, level = 0, evals = 0;
    Fit star;
    int n_in = 0;
};

// RANSAC4RT on host arrays.  hi (nullable): the device's upper bounds for the 500 hypotheses of the first level; idx0 (nullable):
// their sample indices; rnd (nullable when idx0 is given and no level beyond the first is needed): [3 * 500][4] uniform draws.
// -> 0 ok, 1 the draws are needed (an escalation without rnd), -1 LAPACK failure
inline void sample_indices(const double *rnd, const int32_t (*idx_l)[4], int level, int t, int64_t n, int32_t idx[4]) {
    if (idx_l) {   // (the level's indices as the device recorded them)
        for (int q = 0; q < 4; ++q) idx[q] = idx_l[t][q];
    } else {
        const double *r4 = rnd + ((size_t)level * CAELO_RANSAC_MAX_TRIALS + (size_t)t) * 4;
        for (int q = 0; q < 4; ++q) idx[q] = (int32_t)(r4[q] * (double)n);  // :182-184 int32(u * N), with replacement
    }
}

// bounds and sample indices per level (null: none for that level)
struct LevelBounds {
    const int32_t *hi[CAELO_RANSAC_LEVELS] = {nullptr, nullptr, nullptr};
    const int32_t (*idx[CAELO_RANSAC_LEVELS])[4] = {nullptr, nullptr, nullptr};
};

int ransac_host(const float *P0, const float *P1, int64_t n, const double *rnd, LevelBounds lb, Scratch &S,
                Outcome *o, uint8_t *mask_out) {
    S.size(n > 0 ? n : 1);
    const int least = (100 < (int)(0.2 * (double)n)) ? 100 : (int)(0.2 * (double)n);  // :166
    const double min_success = 0.25 * (double)n;                                        // :167
    Outcome out;
    out.star.credible = 1;
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(out.star.R, I, sizeof(I));
    out.star.T[0] = out.star.T[1] = out.star.T[2] = 0.f;
    if (mask_out) memset(mask_out, 0, (size_t)(n > 0 ? n : 0));
    if (n <= 0) { *o = out; return 0; }  // a frame without key points: the pose fails as a value (the kernels' convention)
    float thr = 0.4f;                    // :171 (compared in float32: a Python float beside a float32 array)
    for (int level = 0; level < CAELO_RANSAC_LEVELS; ++level, thr *= 2.0f) {  // :207-214
        out.level = level;
        int32_t *c = S.cnt.data();
        uint8_t *ex = S.exact.data();
        int w = -1, it = 0;
        Fit f;
        const int32_t *hi = lb.hi[level];
        const int32_t (*idx0)[4] = lb.idx[level];
        if (hi != nullptr) {
            // ---- with the device's bounds: c[t] >= the count the reference's arithmetic gives hypothesis t.  Replay the rules over
            // the bounds; the winner of that replay is evaluated exactly; repeat until the winner is exact.  Then every other
            // hypothesis of the replayed range has a count <= its bound <= the winner's (strictly below it in front of the winner),
            // and no bound in front of the exit reached 0.25 N: the replay over the TRUE counts ends at the same iteration with the
            // same winner.
            if (!idx0 && !rnd) return 1;
            for (int t = 0; t < CAELO_RANSAC_MAX_TRIALS; ++t) { c[t] = hi[t]; ex[t] = 0; }
            int held = -1;  // the hypothesis whose mask / pose S.m_best / f hold
            bool violated = false;
            for (;;) {
                replay(c, n, &w, &it);
                if (w < 0 || ex[w]) break;
                int32_t idx[4];
                sample_indices(rnd, idx0, level, w, n, idx);
                int cnt = 0;
                if (!hypothesis(P0, P1, n, idx, thr, S, &f, S.m_best.data(), &cnt)) return -1;
                ++out.evals;
                // The argument above stands on hi[t] >= the reference's count.  The bound's constant is calibrated, not proven
                // (match.hip, hypothesis_bound): every exact count this loop sees is checked against it, and ONE count above its bound
                // discards the bounds of the pair -- the level is redone as the reference's loop stands -- and is counted
                // (caelo_host_bound_violations; the parity soak and bench.py print the counter).
                if (cnt > hi[w]) { violated = true; break; }
                c[w] = cnt;
                ex[w] = 1;
                held = w;
            }
            if (violated) {
                g_bound_violations.fetch_add(1);
                lb.hi[level] = nullptr;
                --level;
                thr *= 0.5f;   // (exact: the loop header doubles it again)
                continue;
            }
            if (w >= 0 && held != w) {  // the winner was evaluated before another candidate: once more for its mask
                int32_t idx[4];
                sample_indices(rnd, idx0, level, w, n, idx);
                int cnt = 0;
                if (!hypothesis(P0, P1, n, idx, thr, S, &f, S.m_best.data(), &cnt)) return -1;
                ++out.evals;
            }
            if (w >= 0) out.star = f;
        } else {
            // ---- without bounds: the reference's loop as it stands (:181-206)
            if (!rnd && !idx0) return 1;
            int best = 0;
            while (it < 100 || (it < CAELO_RANSAC_MAX_TRIALS && (double)best < min_success)) {
                int32_t idx[4];
                sample_indices(rnd, idx0, level, it, n, idx);
                int cnt = 0;
                if (!hypothesis(P0, P1, n, idx, thr, S, &f, S.m_try.data(), &cnt)) return -1;
                ++out.evals;
                c[it] = cnt;
                if (cnt >= least && cnt > best) {  // :195-203
                    best = cnt;
                    w = it;
                    S.m_best.swap(S.m_try);
                    out.star = f;
                }
                ++it;
            }
        }
        out.iterations = it;
        if (w >= 0) {  // the level succeeded with hypothesis w
            out.success = 1;
            out.best_trial = level * CAELO_RANSAC_MAX_TRIALS + w;
            out.n_in = c[w];
            if (mask_out) memcpy(mask_out, S.m_best.data(), (size_t)n);
            *o = out;
            return 0;
        }
        if (least <= 0) {  // N < 5: every hypothesis is admissible and `isSuccess` is set without any inlier (:195-205); identity pose
            out.success = 1;
            *o = out;
            return 0;
        }
    }
    out.level = CAELO_RANSAC_LEVELS - 1;  // :210-212: the threshold is halved back after the last doubling
    *o = out;
    return 0;
}

// SolveRelativePose's tail (Match.py:269-283): the refit over all inliers, same statements
bool refit(const float *P0, const float *P1, int64_t n, const uint8_t *mask, Scratch &S, Fit *f) {
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i)
        if (mask[i]) {
            for (int a = 0; a < 3; ++a) { S.s0[3 * m + a] = P0[3 * i + a]; S.s1[3 * m + a] = P1[3 * i + a]; }
            ++m;
        }
    if (m == 0) return true;  // :275-276: the RANSAC pose is returned as it is
    return solve_rt_rows(S.s0.data(), S.s1.data(), m, S.c0.data(), S.c1.data(), f);
}

void fill_result(const Outcome &o, const Fit &final_fit, int64_t n, caelo_pose_result *r) {
    for (int q = 0; q < 9; ++q) { r->R[q] = final_fit.R[q]; r->R_ransac[q] = o.star.R[q]; }
    for (int q = 0; q < 3; ++q) { r->T[q] = final_fit.T[q]; r->T_ransac[q] = o.star.T[q]; }
    r->threshold = 0.4f * (float)(1 << o.level);
    r->success = o.success;
    r->iterations = o.iterations;
    r->n_inliers = o.n_in;
    r->best_trial = o.best_trial;
    r->n_pairs = (int32_t)n;
}

int certify_one(const float *P0, const float *P1, int64_t n, const double *rnd, const LevelBounds &lb, Scratch &S,
                caelo_pose_result *res, uint8_t *mask, int32_t *evals) {
    Outcome o;
    const int rc = ransac_host(P0, P1, n, rnd, lb, S, &o, mask);
    if (rc != 0) return rc;
    Fit fin = o.star;
    if (o.success && o.n_in > 0 && !refit(P0, P1, n, mask, S, &fin)) return -1;
    fill_result(o, fin, n, res);
    if (evals) *evals = o.evals;
    return 0;
}

}  // namespace

// one record (caelo_pipeline's certifier thread, caelo_host_certify): -> 0 exact, 1 the draws are needed, 2 no bounds in the record,
// 3 no record, -1 LAPACK failure / no BLAS bound
int certify_record(const caelo_ransac_cert &c, const double *rnd, caelo_pose_result *res, uint8_t *mask, int64_t mask_len, int32_t *evals) {
    static thread_local Scratch S;
    if (!g_blas.sgemm) return -1;
    if (c.magic != CAELO_CERT_MAGIC) return 3;
    if (c.n_pairs < 0 || c.n_pairs > CAELO_CERT_MAX_PAIRS || (c.flags & CAELO_CERT_NO_BOUNDS) || c.n_pairs > mask_len) return 2;
    LevelBounds lb;
    lb.hi[0] = c.hi; lb.idx[0] = c.idx;
    if (c.levels_up == 1)   // the kernels saw the first level fail and left the bounds of the 0.8 m / 1.6 m levels (k_ransac_hyp_up)
        for (int l = 1; l < CAELO_RANSAC_LEVELS; ++l) { lb.hi[l] = c.hi_up[l - 1]; lb.idx[l] = c.idx_up[l - 1]; }
    const int rc = certify_one(&c.p0[0][0], &c.p1[0][0], c.n_pairs, rnd, lb, S, res, mask, evals);
    if (rc == 0 && c.n_pairs < mask_len) memset(mask + c.n_pairs, 0, (size_t)(mask_len - c.n_pairs));
    return rc;
}

CAELO_API int caelo_host_bind_blas(void *cblas_sgemm, void *cblas_sgemv, void *dgesdd, int ilp64) {
    CAELO_REQUIRE(cblas_sgemm && cblas_sgemv && dgesdd, "null BLAS entry point");
    g_blas.sgemm = cblas_sgemm; g_blas.sgemv = cblas_sgemv; g_blas.dgesdd = dgesdd; g_blas.ilp64 = ilp64 ? 1 : 0;
    const float H[9] = {2, 0, 0, 0, 1, 0, 0, 0, 0.5f};
    float U[9], V[9];
    if (!svd3(H, U, V, 1)) {
        g_blas = HostBlas();
        CAELO_REQUIRE(false, "dgesdd workspace query failed (wrong integer width?)");
    }
    return CAELO_OK;
}

CAELO_API int caelo_host_blas_bound(void) { return g_blas.sgemm != nullptr ? 1 : 0; }

// The five BLAS / LAPACK calls of the host half, one at a time, exactly as solve_rt_rows / score issue them (same transposition
// flags and leading dimensions): what caelo/hostblas.py compares with np.dot / np.linalg.svd before it accepts a binding.
//   0: out[9]  = np.dot(a.T, b)            a, b [n][3]          (cblas_sgemm Trans, NoTrans, k = n)
//   1: out[18] = U | Vh of np.linalg.svd(a), a [3][3]           (dgesdd on a float64 copy, cast back)
//   2: out[9]  = np.dot(a.T, b.T)          a, b [3][3]          (cblas_sgemm Trans, Trans)
//   3: out[3]  = np.dot(a, b)              a [3][3], b [3]      (cblas_sgemv)
//   4: out[3n] = np.dot(a, b.T)            a [3][3], b [n][3]   (cblas_sgemm NoTrans, Trans)
CAELO_API int caelo_host_blas_probe(int op, const float *a_host, const float *b_host, int64_t n, float *out_host) {
    CAELO_REQUIRE(g_blas.sgemm, "caelo_host_bind_blas was not called");
    CAELO_REQUIRE(a_host && out_host && (b_host || op == 1) && op >= 0 && op <= 4 && n > 0, "bad argument");
    switch (op) {
    case 0: gemm(kTrans, kNoTrans, 3, 3, n, a_host, 3, b_host, 3, out_host, 3); break;
    case 1: CAELO_REQUIRE(svd3(a_host, out_host, out_host + 9), "dgesdd failed"); break;
    case 2: gemm(kTrans, kTrans, 3, 3, 3, a_host, 3, b_host, 3, out_host, 3); break;
    case 3: gemv3(a_host, b_host, out_host); break;
    default: gemm(kNoTrans, kTrans, 3, n, 3, a_host, 3, b_host, 3, out_host, n); break;
    }
    return CAELO_OK;
}

// forget the bound entry points (caelo/hostblas.py: a candidate library that failed the bit-for-bit verification must not stay bound)
CAELO_API int caelo_host_unbind_blas(void) {
    g_blas = HostBlas();
    return CAELO_OK;
}

// exact counts found ABOVE the device's upper bound since the process started (0 on every run so far; a pair where it happens is
// decided by the reference's loop without bounds, so its result is still exact)
CAELO_API int64_t caelo_host_bound_violations(void) { return g_bound_violations.load(); }

CAELO_API int caelo_host_solve_rt(const float *p0_host, const float *p1_host, int64_t n, float *R_host, float *T_host, int32_t *credible_host) {
    CAELO_REQUIRE(g_blas.sgemm, "caelo_host_bind_blas was not called");
    CAELO_REQUIRE(p0_host && p1_host && R_host && T_host && n > 0, "bad argument");
    std::vector<float> c0((size_t)3 * n), c1((size_t)3 * n);
    Fit f;
    CAELO_REQUIRE(solve_rt_rows(p0_host, p1_host, n, c0.data(), c1.data(), &f), "dgesdd failed");
    memcpy(R_host, f.R, sizeof(f.R));
    memcpy(T_host, f.T, sizeof(f.T));
    if (credible_host) *credible_host = f.credible;
    return CAELO_OK;
}

CAELO_API int caelo_host_ransac(const float *pairs0_host, const float *pairs1_host, int64_t n, const double *rand_host, const int32_t *hi_host,
                                caelo_pose_result *result_host, uint8_t *mask_host, int32_t *evals_host) {
    CAELO_REQUIRE(g_blas.sgemm, "caelo_host_bind_blas was not called");
    CAELO_REQUIRE(pairs0_host && pairs1_host && rand_host && result_host && mask_host && n >= 0, "bad argument");
    Scratch S;
    LevelBounds lb;
    lb.hi[0] = hi_host;
    const int rc = certify_one(pairs0_host, pairs1_host, n, rand_host, lb, S, result_host, mask_host, evals_host);
    CAELO_REQUIRE(rc == 0, "dgesdd failed");
    return CAELO_OK;
}

CAELO_API int64_t caelo_cert_bytes(void) { return (int64_t)sizeof(caelo_ransac_cert); }

CAELO_API int caelo_host_certify(const void *certs_host, int64_t k, const double *const *rand_host, caelo_pose_result *results_host,
                                 uint8_t *masks_host, int64_t mask_ld, int32_t *evals_host, int32_t *status_host, int threads) {
    CAELO_REQUIRE(g_blas.sgemm, "caelo_host_bind_blas was not called");
    CAELO_REQUIRE(certs_host && results_host && masks_host && k >= 0 && mask_ld >= CAELO_MAX_KEYPTS, "bad argument");
    const caelo_ransac_cert *certs = (const caelo_ransac_cert *)certs_host;
    std::atomic<int64_t> next(0);
    std::atomic<int> failed(0);
    auto work = [&]() {
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= k) break;
            const int st = certify_record(certs[i], rand_host ? rand_host[i] : nullptr, results_host + i, masks_host + i * mask_ld, mask_ld,
                                          evals_host ? evals_host + i : nullptr);
            if (st < 0) failed.store(1);
            if (status_host) status_host[i] = st < 0 ? 0 : st;
        }
    };
    int nt = threads > 0 ? threads : 1;
    if (nt > k) nt = (int)(k > 0 ? k : 1);
    if (nt <= 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t) pool.emplace_back(work);
        for (auto &t : pool) t.join();
    }
    CAELO_REQUIRE(!failed.load(), "dgesdd failed");
    return CAELO_OK;
}
