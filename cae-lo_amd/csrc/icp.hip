// icp.hip -- the re-registration that follows the odometry, device resident.
//
// Reference behaviour restated here (never its code):
//   ICP                      MyICP.py:26-72     point-to-point on the extended key points
//   GetPtsInliners           MyICP.py:75-85     nearest neighbour (sklearn, exact f64 Euclidean) + distance gate
//   GetPlanarPtsInliners     MyICP.py:88-114    the same on planar points, then the foot of the perpendicular onto the
//                                               plane through the frame-1 point (its stored normal) and a second gate
//   ICP_Pt2PtAndPt2Plane     MyICP.py:127-201   both pair sets in one SolveRT per iteration (caller: RefinePoses.py:291-295)
// The whole iteration loop runs on the device: per iteration one nearest-neighbour launch per point set and one
// single-workgroup update (SolveRT over the pairs, move the frame-1 sets, accumulate R_star / T_star in float64, the
// Euler-angle stop rule and the threshold decay of the reference's loop).  Every launch starts by reading a `done`
// word of the state record, so after the loop has ended the remaining launches fall through; the host reads the
// record once.  Round 1 synchronised twice per iteration (Python loop around caelo_icp_step).
#include "caelo_internal.h"
#include "caelo_rigid.h"

#define ICP_TILE 1024

struct IcpState {
    double R_star[9], T_star[3];
    double thr0, thr1;
    int32_t iter;       // iterations completed
    int32_t done;       // the loop has ended
    int32_t success;
    int32_t n_pts, n_planar;  // pairs of the last evaluated iteration
    int32_t cnt_pts, cnt_planar;  // scratch: inlier counters of the iteration in flight
    int32_t pad;
};
static_assert(sizeof(IcpState) <= sizeof(caelo_icp_result) + 64, "state record layout");

// nearest neighbour in set 0 (stride ld0) of every point of set 1 (stride ld1); pairs closer than the threshold of the
// state record (which = 0: thr0, 1: thr1).  Exact f64 distance, first minimum, like sklearn's kd-tree on these inputs.
__global__ void __launch_bounds__(256) k_icp_nn2(const float *__restrict__ p0, int ld0, int n0, const float *__restrict__ p1, int ld1, int n1,
                                                 IcpState *st, int which, int64_t *__restrict__ idx0, uint8_t *__restrict__ mask) {
    if (st->done) return;
    __shared__ float tile[ICP_TILE * 3];
    const int tid = threadIdx.x;
    const int j = blockIdx.x * blockDim.x + tid;
    const double thr = which ? st->thr1 : st->thr0;
    double qx = 0.0, qy = 0.0, qz = 0.0;
    if (j < n1) { qx = p1[(size_t)ld1 * j]; qy = p1[(size_t)ld1 * j + 1]; qz = p1[(size_t)ld1 * j + 2]; }
    double best = 1.0e300;
    int besti = 0;
    for (int base = 0; base < n0; base += ICP_TILE) {
        const int m = min(ICP_TILE, n0 - base);
        __syncthreads();
        for (int i = tid; i < m; i += 256) {
            const float *s = p0 + (size_t)ld0 * (base + i);
            tile[3 * i] = s[0]; tile[3 * i + 1] = s[1]; tile[3 * i + 2] = s[2];
        }
        __syncthreads();
        for (int i = 0; i < m; ++i) {  // all lanes read the same address: LDS broadcast
            const double dx = (double)tile[3 * i] - qx, dy = (double)tile[3 * i + 1] - qy, dz = (double)tile[3 * i + 2] - qz;
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < best) { best = d2; besti = base + i; }
        }
    }
    if (j < n1) {
        idx0[j] = besti;
        mask[j] = (n0 > 0 && sqrt(best) < thr) ? 1 : 0;  // distances < inlierThreshold (:80, :100)
    }
}

// planar pair of frame-1 planar point j (MyICP.py:103-113): returns false when it is gated out
__device__ inline bool planar_pair(const float *pn0, const float *pn1, const int64_t *idx0, const uint8_t *mask, int j, double thr0,
                                   float pedal[3], float in1[3]) {
    if (!mask[j]) return false;
    const float *a = pn0 + 6 * (size_t)idx0[j];  // inliers0 (xyz of the frame-0 neighbour)
    const float *b = pn1 + 6 * (size_t)j;        // inliers1 | norms1
    const float v0 = __fsub_rn(a[0], b[0]), v1 = __fsub_rn(a[1], b[1]), v2 = __fsub_rn(a[2], b[2]);                    // :104
    const float d = __fadd_rn(__fadd_rn(__fmul_rn(b[3], v0), __fmul_rn(b[4], v1)), __fmul_rn(b[5], v2));              // :105
    pedal[0] = __fadd_rn(b[0], __fmul_rn(b[3], d)); pedal[1] = __fadd_rn(b[1], __fmul_rn(b[4], d)); pedal[2] = __fadd_rn(b[2], __fmul_rn(b[5], d));  // :106
    const float e0 = __fsub_rn(pedal[0], b[0]), e1 = __fsub_rn(pedal[1], b[1]), e2 = __fsub_rn(pedal[2], b[2]);
    const float dist = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(e0, e0), __fmul_rn(e1, e1)), __fmul_rn(e2, e2)));          // :108
    in1[0] = b[0]; in1[1] = b[1]; in1[2] = b[2];
    return (double)dist < thr0;                                                                                         // :109
}

#define ICP_TERMS 16
struct IcpParams {
    int32_t max_iter, min_iter, min_pairs, fail_only_first;  // fail_only_first: too few pairs is a failure only at iteration 0 (:166-169)
    double decay0, decay1, small_shift, ep;
    int32_t planar;  // the planar sets take part
    int32_t pad;
};

__global__ void __launch_bounds__(256) k_icp_update(const float *__restrict__ pc0, float *__restrict__ pc1, int n1, const int64_t *__restrict__ idx_p,
                                                    const uint8_t *__restrict__ mask_p, const float *__restrict__ pn0, float *__restrict__ pn1, int m1,
                                                    const int64_t *__restrict__ idx_q, const uint8_t *__restrict__ mask_q, IcpState *st, IcpParams prm) {
    if (st->done) return;
    __shared__ double red[4][ICP_TERMS + 2];
    __shared__ float s_rt[12];
    __shared__ int s_stop;
    const int tid = threadIdx.x;
    const double thr0 = st->thr0;
    const bool use_pts = st->iter < 100;  // :146-152 (the loop never gets there with the reference's maxIterTimes = 50)
    double a[ICP_TERMS + 2];
#pragma unroll
    for (int t = 0; t < ICP_TERMS + 2; ++t) a[t] = 0.0;
#define ICP_ACC(X0, Y0, Z0, X1, Y1, Z1)                                               \
    {                                                                                 \
        const double x0 = X0, y0 = Y0, z0 = Z0, x1 = X1, y1 = Y1, z1 = Z1;            \
        a[0] += 1.0;                                                                  \
        a[1] += x0; a[2] += y0; a[3] += z0; a[4] += x1; a[5] += y1; a[6] += z1;       \
        a[7] += x1 * x0; a[8] += x1 * y0; a[9] += x1 * z0;                            \
        a[10] += y1 * x0; a[11] += y1 * y0; a[12] += y1 * z0;                         \
        a[13] += z1 * x0; a[14] += z1 * y0; a[15] += z1 * z0;                         \
    }
    for (int i = tid; i < n1; i += 256) {
        if (!mask_p[i]) continue;
        a[16] += 1.0;
        if (!use_pts) continue;
        const float *u = pc0 + 3 * (size_t)idx_p[i], *v = pc1 + 3 * (size_t)i;
        ICP_ACC(u[0], u[1], u[2], v[0], v[1], v[2])
    }
    if (prm.planar) {
        for (int j = tid; j < m1; j += 256) {
            float pedal[3], in1[3];
            if (!planar_pair(pn0, pn1, idx_q, mask_q, j, thr0, pedal, in1)) continue;
            a[17] += 1.0;
            ICP_ACC(pedal[0], pedal[1], pedal[2], in1[0], in1[1], in1[2])
        }
    }
#pragma unroll
    for (int t = 0; t < ICP_TERMS + 2; ++t)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a[t] += __shfl_xor(a[t], o);
    if ((tid & 63) == 0)
#pragma unroll
        for (int t = 0; t < ICP_TERMS + 2; ++t) red[tid >> 6][t] = a[t];
    __syncthreads();
    if (tid == 0) {
        double s[ICP_TERMS + 2];
        for (int t = 0; t < ICP_TERMS + 2; ++t) s[t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
        const int pairs = (int)s[0];
        st->n_pts = (int)s[16];
        st->n_planar = (int)s[17];
        int stop = 0;
        if (pairs < prm.min_pairs) {  // ICP :38-40 / Pt2Plane :166-169
            if (!prm.fail_only_first || st->iter < 1) st->success = 0;
            stop = 1;
        } else {
            const double cnt = s[0];
            const double m0[3] = {s[1] / cnt, s[2] / cnt, s[3] / cnt}, m1c[3] = {s[4] / cnt, s[5] / cnt, s[6] / cnt};
            double H[9];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) H[3 * i + j] = s[7 + 3 * i + j] - cnt * m1c[i] * m0[j];
            float R[9], T[3];
            rigid_from_H(H, m0, m1c, R, T);
            for (int i = 0; i < 9; ++i) s_rt[i] = R[i];
            for (int i = 0; i < 3; ++i) s_rt[9 + i] = T[i];
            // R_star = R R_star, T_star = R T_star + T in float64 (:50-51)
            double Rn[9], Tn[3];
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j)
                    Rn[3 * i + j] = (double)R[3 * i] * st->R_star[j] + (double)R[3 * i + 1] * st->R_star[3 + j] + (double)R[3 * i + 2] * st->R_star[6 + j];
                Tn[i] = (double)R[3 * i] * st->T_star[0] + (double)R[3 * i + 1] * st->T_star[1] + (double)R[3 * i + 2] * st->T_star[2] + (double)T[i];
            }
            for (int i = 0; i < 9; ++i) st->R_star[i] = Rn[i];
            for (int i = 0; i < 3; ++i) st->T_star[i] = Tn[i];
            // stop rule and threshold decay (:54-65): Euler angles in degrees (Transformations.py:181-186), norm of T in float32
            const double r2d = 180.0 / 3.14159265358979323846;
            const double e0 = atan2((double)R[7], (double)R[8]) * r2d;
            const double e1 = atan2(-(double)R[6], sqrt((double)R[7] * (double)R[7] + (double)R[8] * (double)R[8])) * r2d;
            const double e2 = atan2((double)R[3], (double)R[0]) * r2d;
            const double normE = sqrt(e0 * e0 + e1 * e1 + e2 * e2);
            const double normT = (double)sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(T[0], T[0]), __fmul_rn(T[1], T[1])), __fmul_rn(T[2], T[2])));
            const int it = st->iter;
            if (it >= prm.min_iter && normE < prm.ep && normT < prm.ep) stop = 2;  // converged: this iteration's move still applies
            else if (normE < prm.small_shift && normT < prm.small_shift) { st->thr0 *= prm.decay0; st->thr1 *= prm.decay1; }
            st->iter = it + 1;
            if (it + 1 >= prm.max_iter) stop = stop ? stop : 2;
        }
        s_stop = stop;
        if (stop) st->done = 1;
    }
    __syncthreads();
    if (s_stop == 1) return;  // too few pairs: nothing is moved
    const float r0 = s_rt[0], r1 = s_rt[1], r2 = s_rt[2], r3 = s_rt[3], r4 = s_rt[4], r5 = s_rt[5], r6 = s_rt[6], r7 = s_rt[7], r8 = s_rt[8];
    const float t0 = s_rt[9], t1 = s_rt[10], t2 = s_rt[11];
    for (int j = tid; j < n1; j += 256) {  // PC1 = (np.dot(R, PC1.T) + T).T  (:49)
        const float x = pc1[3 * (size_t)j], y = pc1[3 * (size_t)j + 1], z = pc1[3 * (size_t)j + 2];
        pc1[3 * (size_t)j] = r0 * x + r1 * y + r2 * z + t0;
        pc1[3 * (size_t)j + 1] = r3 * x + r4 * y + r5 * z + t1;
        pc1[3 * (size_t)j + 2] = r6 * x + r7 * y + r8 * z + t2;
    }
    if (prm.planar)
        for (int j = tid; j < m1; j += 256) {  // the planar points move, their normals do not (:177)
            float *q = pn1 + 6 * (size_t)j;
            const float x = q[0], y = q[1], z = q[2];
            q[0] = r0 * x + r1 * y + r2 * z + t0;
            q[1] = r3 * x + r4 * y + r5 * z + t1;
            q[2] = r6 * x + r7 * y + r8 * z + t2;
        }
}

__global__ void k_icp_init(IcpState *st, double thr0, double thr1) {
    for (int i = 0; i < 9; ++i) st->R_star[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 3; ++i) st->T_star[i] = 0.0;
    st->thr0 = thr0; st->thr1 = thr1;
    st->iter = 0; st->done = 0; st->success = 1; st->n_pts = 0; st->n_planar = 0;
}

__global__ void k_icp_result(const IcpState *st, caelo_icp_result *out) {
    for (int i = 0; i < 9; ++i) out->R_star[i] = st->R_star[i];
    for (int i = 0; i < 3; ++i) out->T_star[i] = st->T_star[i];
    out->threshold0 = st->thr0; out->threshold1 = st->thr1;
    out->iterations = st->iter; out->success = st->success; out->n_inliers_pts = st->n_pts; out->n_inliers_planar = st->n_planar;
}

CAELO_API int64_t caelo_icp_loop_ws_bytes(int64_t n1, int64_t m1) {
    return 256 + ((n1 + m1) * 9 + 255) / 256 * 256 + 256;
}

CAELO_API int caelo_icp(caelo_ctx *c, const float *pc0, int64_t n0, float *pc1, int64_t n1, const float *planar0, int64_t m0,
                        float *planar1, int64_t m1, const caelo_icp_params *prm, caelo_icp_result *result, void *ws, void *stream) {
    CAELO_REQUIRE(c && pc0 && pc1 && prm && result && ws, "null argument");
    CAELO_REQUIRE(n0 > 0 && n1 > 0 && n0 < (1 << 30) && n1 < (1 << 30), "bad shape");
    const bool planar = prm->use_planar != 0;
    if (planar) {
        // sklearn's NearestNeighbors.fit refuses an empty set (MyICP.py:94): the reference raises ValueError -- and
        // GetKeyPtsByAE always returns an empty PlanarPts (SphericalRing.py:219,285)
        CAELO_REQUIRE(planar0 && planar1 && m0 > 0 && m1 > 0, "Found array with 0 sample(s) while a minimum of 1 is required (planar points)");
        CAELO_REQUIRE(m0 < (1 << 30) && m1 < (1 << 30), "bad shape");
    }
    CAELO_REQUIRE(prm->max_iter >= 1 && prm->max_iter <= 1000, "max_iter must be in [1, 1000]");
    hipStream_t s = caelo_stream(stream);
    IcpState *st = (IcpState *)ws;
    int64_t *idx_p = (int64_t *)((char *)ws + 256);
    int64_t *idx_q = idx_p + n1;
    uint8_t *mask_p = (uint8_t *)(idx_q + (planar ? m1 : 0));
    uint8_t *mask_q = mask_p + n1;
    IcpParams kp;
    kp.max_iter = prm->max_iter; kp.min_iter = prm->min_iter; kp.min_pairs = prm->min_pairs; kp.fail_only_first = prm->fail_only_first;
    kp.decay0 = prm->decay0; kp.decay1 = prm->decay1; kp.small_shift = prm->small_shift; kp.ep = prm->ep;
    kp.planar = planar ? 1 : 0; kp.pad = 0;
    k_icp_init<<<1, 1, 0, s>>>(st, prm->threshold0, prm->threshold1);
    CAELO_LAUNCH_CHECK();
    for (int it = 0; it < prm->max_iter; ++it) {
        k_icp_nn2<<<(unsigned)((n1 + 255) / 256), 256, 0, s>>>(pc0, 3, (int)n0, pc1, 3, (int)n1, st, 0, idx_p, mask_p);
        CAELO_LAUNCH_CHECK();
        if (planar) {
            k_icp_nn2<<<(unsigned)((m1 + 255) / 256), 256, 0, s>>>(planar0, 6, (int)m0, planar1, 6, (int)m1, st, 1, idx_q, mask_q);
            CAELO_LAUNCH_CHECK();
        }
        k_icp_update<<<1, 256, 0, s>>>(pc0, pc1, (int)n1, idx_p, mask_p, planar0, planar1, planar ? (int)m1 : 0, idx_q, mask_q, st, kp);
        CAELO_LAUNCH_CHECK();
    }
    k_icp_result<<<1, 1, 0, s>>>(st, result);
    CAELO_LAUNCH_CHECK();
    return CAELO_OK;
}
