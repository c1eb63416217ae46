// caelo_rigid.h -- 3x3 rigid fit (SolveRT, Match.py:138-158) shared by match.hip and icp.hip.  Device code only.
#pragma once
#include "caelo_internal.h"

// ------------------------------------------------------------------------------------------------
// 3x3 rigid fit from a cross-covariance H = sum (p1 - m1)(p0 - m0)^T   (Match.py:141-157)
// one-sided Jacobi SVD in f64: H V = U S ; R = V U^T (the reference's V.T @ U.T with V = Vh);
// det(R) < 0 -> the reference negates column 2 of Vh, i.e. R <- diag(1,1,-1) R  (:151-155).
// ------------------------------------------------------------------------------------------------
// Written on scalars only (every index a compile-time constant after unrolling): no private arrays that would live
// in scratch memory -- none of the pair kernels uses any.
#define JAC_ROT(AP, AQ, VP, VQ)                                       \
    {                                                                 \
        const double ap_ = AP, aq_ = AQ;                              \
        AP = cs * ap_ - sn * aq_;                                     \
        AQ = sn * ap_ + cs * aq_;                                     \
        const double vp_ = VP, vq_ = VQ;                              \
        VP = cs * vp_ - sn * vq_;                                     \
        VQ = sn * vp_ + cs * vq_;                                     \
    }
// one Jacobi rotation of columns p, q (given as their three entries of A and V); returns |gamma| or 0 when skipped
#define JAC_PAIR(A0P, A1P, A2P, A0Q, A1Q, A2Q, V0P, V1P, V2P, V0Q, V1Q, V2Q)                                 \
    {                                                                                                        \
        double alpha = 0, beta = 0, gamma = 0;                                                               \
        alpha += A0P * A0P; beta += A0Q * A0Q; gamma += A0P * A0Q;                                           \
        alpha += A1P * A1P; beta += A1Q * A1Q; gamma += A1P * A1Q;                                           \
        alpha += A2P * A2P; beta += A2Q * A2Q; gamma += A2P * A2Q;                                           \
        const double lim = 1e-30 + 1e-16 * sqrt(alpha * beta);                                               \
        if (!(fabs(gamma) <= lim)) {                                                                         \
            offmax = fmax(offmax, fabs(gamma));                                                              \
            const double zeta = (beta - alpha) / (2.0 * gamma);                                              \
            const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));              \
            const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;                                          \
            JAC_ROT(A0P, A0Q, V0P, V0Q) JAC_ROT(A1P, A1Q, V1P, V1Q) JAC_ROT(A2P, A2Q, V2P, V2Q)              \
        }                                                                                                    \
    }
#define JAC_SWAP(X, Y) { const double t_ = X; X = Y; Y = t_; }
// SIGN3 = false: the rank-2 completion is right-handed whatever the sign of det H (the RANSAC hypothesis kernels: three live doubles
// at their register peak; every rank-2 sample is a kind-1 hypothesis of the certificate there, both poses scored -- match.hip)
template <bool SIGN3 = true>
__device__ inline int rigid_from_H_jacobi(const double Hin[9], const double m0[3], const double m1[3], float R[9], float T[3]) {
    // A = H (columns 0, 1, 2 as a*0, a*1, a*2), V = I
    double a00 = Hin[0], a01 = Hin[1], a02 = Hin[2], a10 = Hin[3], a11 = Hin[4], a12 = Hin[5], a20 = Hin[6], a21 = Hin[7], a22 = Hin[8];
    double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
#pragma unroll 1
    for (int sweep = 0; sweep < 12; ++sweep) {
        double offmax = 0.0;
        JAC_PAIR(a00, a10, a20, a01, a11, a21, v00, v10, v20, v01, v11, v21)  // (p, q) = (0, 1)
        JAC_PAIR(a00, a10, a20, a02, a12, a22, v00, v10, v20, v02, v12, v22)  // (0, 2)
        JAC_PAIR(a01, a11, a21, a02, a12, a22, v01, v11, v21, v02, v12, v22)  // (1, 2)
        if (offmax == 0.0) break;
    }
    // columns of A are u_i * s_i; order by descending s so a (near-)null direction ends up last (the same three
    // compare-exchanges as a bubble sort of the column order)
    double s0 = sqrt(a00 * a00 + a10 * a10 + a20 * a20), s1 = sqrt(a01 * a01 + a11 * a11 + a21 * a21), s2 = sqrt(a02 * a02 + a12 * a12 + a22 * a22);
    if (s1 > s0) { JAC_SWAP(s0, s1) JAC_SWAP(a00, a01) JAC_SWAP(a10, a11) JAC_SWAP(a20, a21) JAC_SWAP(v00, v01) JAC_SWAP(v10, v11) JAC_SWAP(v20, v21) }
    if (s2 > s0) { JAC_SWAP(s0, s2) JAC_SWAP(a00, a02) JAC_SWAP(a10, a12) JAC_SWAP(a20, a22) JAC_SWAP(v00, v02) JAC_SWAP(v10, v12) JAC_SWAP(v20, v22) }
    if (s2 > s1) { JAC_SWAP(s1, s2) JAC_SWAP(a01, a02) JAC_SWAP(a11, a12) JAC_SWAP(a21, a22) JAC_SWAP(v01, v02) JAC_SWAP(v11, v12) JAC_SWAP(v21, v22) }
    const double i0 = s0 > 0 ? 1.0 / s0 : 0.0, i1 = s1 > 0 ? 1.0 / s1 : 0.0, i2 = s2 > 0 ? 1.0 / s2 : 0.0;
    // U = [u0 u1 u2] (column j = entries u0j, u1j, u2j), W = the matching columns of V
    double u00 = a00 * i0, u10 = a10 * i0, u20 = a20 * i0, u01 = a01 * i1, u11 = a11 * i1, u21 = a21 * i1, u02 = a02 * i2, u12 = a12 * i2, u22 = a22 * i2;
    const double tiny = 1e-12 * (s0 > 0 ? s0 : 1.0);
    if (s1 <= tiny) {  // rank <= 1: any orthonormal completion (the pose is meaningless anyway)
        double e0 = 1, e1 = 0;
        const double e2 = 0;
        if (fabs(u00) > 0.9) { e0 = 0; e1 = 1; }
        const double d = e0 * u00 + e1 * u10 + e2 * u20;
        const double w0 = e0 - d * u00, w1 = e1 - d * u10, w2 = e2 - d * u20;
        const double nv = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
        u01 = w0 / nv; u11 = w1 / nv; u21 = w2 / nv;
    }
    if (s2 <= tiny) {
        // rank 2 to working precision: u3 = +-(u1 x u2).  The SIGN is the one H v3 = s3 u3 still has while s3 is small but not zero
        // (round 6: a sample of four ground points, s3 / s1 = 1e-13 and det H = +5.7e-3 safely positive, got the right-handed
        // completion whatever the column swaps above had done to det V -- an improper V U^T, the reflection quirk of :151-155, and a
        // pose 4.5e-5 rad away from the reference's, which had read the sign off its SVD: one inlier count above its "upper bound",
        // tests/golden/ransac_bound_case.npz).  With s3 exactly zero the choice is free (the callers that care score both poses:
        // rigid_two_candidates, kind 1 of the certificate).
        // (the triple product first, then the cross product again into the three slots: three live doubles less than keeping it -- this
        //  branch sits at the register peak of k_ransac_hyp, whose 168 registers are what fits beside two stage-1 workgroups)
        double tp = u02 * (u10 * u21 - u20 * u11);   // u.2 = H v3 / s3 as far as it is known; zero when s3 is
        tp += u12 * (u20 * u01 - u00 * u21);
        tp += u22 * (u00 * u11 - u10 * u01);
        const double sgn = (SIGN3 && tp < 0.0) ? -1.0 : 1.0;
        u02 = sgn * (u10 * u21 - u20 * u11);
        u12 = sgn * (u20 * u01 - u00 * u21);
        u22 = sgn * (u00 * u11 - u10 * u01);
    }
    // R = W U^T
    double r0 = v00 * u00 + v01 * u01 + v02 * u02, r1 = v00 * u10 + v01 * u11 + v02 * u12, r2 = v00 * u20 + v01 * u21 + v02 * u22;
    double r3 = v10 * u00 + v11 * u01 + v12 * u02, r4 = v10 * u10 + v11 * u11 + v12 * u12, r5 = v10 * u20 + v11 * u21 + v12 * u22;
    double r6 = v20 * u00 + v21 * u01 + v22 * u02, r7 = v20 * u10 + v21 * u11 + v22 * u12, r8 = v20 * u20 + v21 * u21 + v22 * u22;
    const double det = r0 * (r4 * r8 - r5 * r7) - r1 * (r3 * r8 - r5 * r6) + r2 * (r3 * r7 - r4 * r6);
    if (det < 0) { r6 = -r6; r7 = -r7; r8 = -r8; }  // :151-155
    R[0] = (float)r0; R[1] = (float)r1; R[2] = (float)r2; R[3] = (float)r3; R[4] = (float)r4; R[5] = (float)r5;
    R[6] = (float)r6; R[7] = (float)r7; R[8] = (float)r8;
    T[0] = (float)(m0[0] - (r0 * m1[0] + r1 * m1[1] + r2 * m1[2]));  // :157
    T[1] = (float)(m0[1] - (r3 * m1[0] + r4 * m1[1] + r5 * m1[2]));
    T[2] = (float)(m0[2] - (r6 * m1[0] + r7 * m1[1] + r8 * m1[2]));
    return det < 0 ? -1 : 1;  // isCredible (:139,:152)
}

// The two poses SolveRT can return for a (numerically) RANK-2 covariance H (Match.py:148-157).  With sigma_3 = 0 the third
// singular vectors are fixed up to a sign each, the SVD routine picks them, and the product V U^T comes out proper or
// improper accordingly; the reference then either keeps it or applies its reflection quirk (Vh[:, 2] *= -1, i.e. row 2 of R
// negated).  So the reference returns either Ra = w1 u1^T + w2 u2^T + w3 u3^T (u3 = u1 x u2, w3 = w1 x w2: det +1) or
// Rb = diag(1, 1, -1) (Ra - 2 w3 u3^T).  Both are built here from the one-sided Jacobi SVD; the caller scores both.
// Returns false when sigma_2 vanishes too (rank <= 1: a one-parameter family of poses, no finite list of candidates).
__device__ inline bool rigid_two_candidates(const double Hin[9], const double m0[3], const double m1[3], float Ra[9], float Ta[3], float Rb[9],
                                            float Tb[3]) {
    double a00 = Hin[0], a01 = Hin[1], a02 = Hin[2], a10 = Hin[3], a11 = Hin[4], a12 = Hin[5], a20 = Hin[6], a21 = Hin[7], a22 = Hin[8];
    double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
#pragma unroll 1
    for (int sweep = 0; sweep < 12; ++sweep) {
        double offmax = 0.0;
        JAC_PAIR(a00, a10, a20, a01, a11, a21, v00, v10, v20, v01, v11, v21)
        JAC_PAIR(a00, a10, a20, a02, a12, a22, v00, v10, v20, v02, v12, v22)
        JAC_PAIR(a01, a11, a21, a02, a12, a22, v01, v11, v21, v02, v12, v22)
        if (offmax == 0.0) break;
    }
    double s0 = sqrt(a00 * a00 + a10 * a10 + a20 * a20), s1 = sqrt(a01 * a01 + a11 * a11 + a21 * a21), s2 = sqrt(a02 * a02 + a12 * a12 + a22 * a22);
    if (s1 > s0) { JAC_SWAP(s0, s1) JAC_SWAP(a00, a01) JAC_SWAP(a10, a11) JAC_SWAP(a20, a21) JAC_SWAP(v00, v01) JAC_SWAP(v10, v11) JAC_SWAP(v20, v21) }
    if (s2 > s0) { JAC_SWAP(s0, s2) JAC_SWAP(a00, a02) JAC_SWAP(a10, a12) JAC_SWAP(a20, a22) JAC_SWAP(v00, v02) JAC_SWAP(v10, v12) JAC_SWAP(v20, v22) }
    if (s2 > s1) { JAC_SWAP(s1, s2) JAC_SWAP(a01, a02) JAC_SWAP(a11, a12) JAC_SWAP(a21, a22) JAC_SWAP(v01, v02) JAC_SWAP(v11, v12) JAC_SWAP(v21, v22) }
    if (!(s1 > 1e-7 * s0) || !(s0 > 0.0)) return false;
    const double i0 = 1.0 / s0, i1 = 1.0 / s1;
    const double u00 = a00 * i0, u10 = a10 * i0, u20 = a20 * i0, u01 = a01 * i1, u11 = a11 * i1, u21 = a21 * i1;
    // third vectors by orthogonality (right-handed both): u3 = u1 x u2, w3 = w1 x w2
    const double u02 = u10 * u21 - u20 * u11, u12 = u20 * u01 - u00 * u21, u22 = u00 * u11 - u10 * u01;
    const double w02 = v10 * v21 - v20 * v11, w12 = v20 * v01 - v00 * v21, w22 = v00 * v11 - v10 * v01;
    // Ra = W U^T with those columns
    const double r0 = v00 * u00 + v01 * u01 + w02 * u02, r1 = v00 * u10 + v01 * u11 + w02 * u12, r2 = v00 * u20 + v01 * u21 + w02 * u22;
    const double r3 = v10 * u00 + v11 * u01 + w12 * u02, r4 = v10 * u10 + v11 * u11 + w12 * u12, r5 = v10 * u20 + v11 * u21 + w12 * u22;
    const double r6 = v20 * u00 + v21 * u01 + w22 * u02, r7 = v20 * u10 + v21 * u11 + w22 * u12, r8 = v20 * u20 + v21 * u21 + w22 * u22;
    // Rb = D (Ra - 2 w3 u3^T), D = diag(1, 1, -1)
    const double q0 = r0 - 2.0 * w02 * u02, q1 = r1 - 2.0 * w02 * u12, q2 = r2 - 2.0 * w02 * u22;
    const double q3 = r3 - 2.0 * w12 * u02, q4 = r4 - 2.0 * w12 * u12, q5 = r5 - 2.0 * w12 * u22;
    const double q6 = -(r6 - 2.0 * w22 * u02), q7 = -(r7 - 2.0 * w22 * u12), q8 = -(r8 - 2.0 * w22 * u22);
    Ra[0] = (float)r0; Ra[1] = (float)r1; Ra[2] = (float)r2; Ra[3] = (float)r3; Ra[4] = (float)r4; Ra[5] = (float)r5; Ra[6] = (float)r6; Ra[7] = (float)r7; Ra[8] = (float)r8;
    Rb[0] = (float)q0; Rb[1] = (float)q1; Rb[2] = (float)q2; Rb[3] = (float)q3; Rb[4] = (float)q4; Rb[5] = (float)q5; Rb[6] = (float)q6; Rb[7] = (float)q7; Rb[8] = (float)q8;
    Ta[0] = (float)(m0[0] - (r0 * m1[0] + r1 * m1[1] + r2 * m1[2]));
    Ta[1] = (float)(m0[1] - (r3 * m1[0] + r4 * m1[1] + r5 * m1[2]));
    Ta[2] = (float)(m0[2] - (r6 * m1[0] + r7 * m1[1] + r8 * m1[2]));
    Tb[0] = (float)(m0[0] - (q0 * m1[0] + q1 * m1[1] + q2 * m1[2]));
    Tb[1] = (float)(m0[1] - (q3 * m1[0] + q4 * m1[1] + q5 * m1[2]));
    Tb[2] = (float)(m0[2] - (q6 * m1[0] + q7 * m1[1] + q8 * m1[2]));
    return true;
}

// Fast path: R = V U^T is the orthogonal polar factor of H^T.  Scaled Newton iteration
// X <- (g X + X^-T / g) / 2 (Higham) converges quadratically in f64 (5-7 steps, no sqrt/div chains of a
// Jacobi SVD: ~10x shorter dependency chain, and every hypothesis wavefront runs this serially).
// Rank-deficient or badly conditioned H (repeated sample indices) falls back to the Jacobi SVD above.
template <bool SIGN3 = true>
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
    if (!ok) return rigid_from_H_jacobi<SIGN3>(H, m0, m1, R, T);
    if (det0 < 0) { X[6] = -X[6]; X[7] = -X[7]; X[8] = -X[8]; }  // Match.py:151-155: Vh[:,2] *= -1  <=>  negate row 2 of R
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = (float)X[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        T[i] = (float)(m0[i] - (X[3 * i] * m1[0] + X[3 * i + 1] * m1[1] + X[3 * i + 2] * m1[2]));  // :157
    return det0 < 0 ? -1 : 1;
}

