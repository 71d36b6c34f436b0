/*
 * forster_oracle.c -- CPU restatement of the "Forster discrete" comparator of the reference
 * (GraphSolver::createimufactor_discrete, GraphSolver_IMU.cpp:141-232, swapcovariance :240-254).
 *
 * TEST INFRASTRUCTURE ONLY (see cpi_oracle.h).
 *
 * PARITY UNPINNED.  The arithmetic of this path lives entirely in a third-party dependency that is
 * absent from /root/reference and from this image: GTSAM, pinned by the reference's ReadMe.md:43-54 to
 * commit c21186c6212798e665da6b5015296713ddfe8c1d, built with -DGTSAM_TANGENT_PREINTEGRATION=OFF, i.e.
 *     gtsam::PreintegratedCombinedMeasurements  over  gtsam::ManifoldPreintegration
 * (on-manifold preintegration of Forster, Carlone, Dellaert, Scaramuzza, RSS 2015 / T-RO 2017, with the
 * bias-augmented 15x15 covariance of Carlone et al.).  What follows restates that published algorithm in
 * GTSAM's own formulation and state order -- NavState::update / NavState::retract chain-rule Jacobians,
 * ManifoldPreintegration::update bias Jacobians, CombinedImuFactor's F P F^T + G covariance step as dense
 * 15x15 products in the order [R p v b_a b_g] -- and then applies the reference's call-site conversions
 * (transpose / rot_2_quat of deltaRij, sign of delRdelBiasOmega, the 1<->4 block swap).  It is anchored
 * on the reference's call site only; no GTSAM output exists here to pin it against.  What the tests pin
 * instead (tests/test_forster_oracle.py): A, B, C are the true derivatives of the discrete update in
 * NavState's local coordinates (finite differences), the bias Jacobians are the true derivatives of the
 * preintegrated means with respect to the bias (finite differences of the integrator itself), and the
 * covariance converges to the continuous CPI covariance (pinned to the compiled reference) as dt -> 0.
 *
 * Documented deviation: an interval with dt == 0 makes GTSAM divide by dt (NaN covariance); here, and in
 * the HIP kernel, such an interval is skipped, as CpiV1.h:72-74 does.  (dt < 0 is skipped by the caller,
 * GraphSolver_IMU.cpp:171.)
 *
 * Internally matrices are row-major.
 */
#include "cpi_oracle.h"
#include <float.h>
#include <math.h>
#include <string.h>

void cpi_oracle_rot_2_quat_rm(const double *rot_rowmajor, double *q); /* cpi_oracle.c (quat_ops.h:45-86) */

static void f_eye(double *A) { memset(A, 0, 9 * sizeof(double)); A[0] = A[4] = A[8] = 1.0; }
static void f_skew(const double *w, double *S) {
    S[0] = 0; S[1] = -w[2]; S[2] = w[1];
    S[3] = w[2]; S[4] = 0; S[5] = -w[0];
    S[6] = -w[1]; S[7] = w[0]; S[8] = 0;
}
static void f_mm(const double *A, const double *B, double *C) {
    double T[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
        T[i * 3 + j] = s;
    }
    memcpy(C, T, sizeof T);
}
static void f_tr(const double *A, double *At) {
    double T[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T[j * 3 + i] = A[i * 3 + j];
    memcpy(At, T, sizeof T);
}
static void f_mv(const double *A, const double *x, double *y) {
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = A[i * 3] * x[0] + A[i * 3 + 1] * x[1] + A[i * 3 + 2] * x[2];
    memcpy(y, t, sizeof t);
}

/* gtsam::SO3::Expmap with its right Jacobian (so3::DexpFunctor): R = I + sin(t) K + (1-cos t) K^2,
 * dexp = I - (1-cos t)/t K + (1 - sin(t)/t) K^2, K = [omega]x / t; first-order forms below sqrt(eps). */
static void so3_expmap(const double *om, double *R, double *H) {
    double W[9], K[9], KK[9];
    const double theta2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    f_skew(om, W);
    f_eye(R);
    if (H) f_eye(H);
    if (theta2 <= DBL_EPSILON) {
        for (int i = 0; i < 9; i++) { R[i] += W[i]; if (H) H[i] -= 0.5 * W[i]; }
        return;
    }
    const double theta = sqrt(theta2);
    for (int i = 0; i < 9; i++) K[i] = W[i] / theta;
    f_mm(K, K, KK);
    const double sin_theta = sin(theta), s2 = sin(theta / 2.0), one_minus_cos = 2.0 * s2 * s2;
    for (int i = 0; i < 9; i++) R[i] += sin_theta * K[i] + one_minus_cos * KK[i];
    if (H) {
        const double a = one_minus_cos / theta, b = 1.0 - sin_theta / theta;
        for (int i = 0; i < 9; i++) H[i] += -a * K[i] + b * KK[i];
    }
}

typedef struct { double R[9], t[3], v[3]; } fs_nav;   /* gtsam::NavState: nRb, n_t, n_v */

static void set9(double *M, int ld, int r, int c, const double *B3, double s) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[(r + i) * ld + c + j] = s * B3[i * 3 + j];
}

/* gtsam::NavState::retract(xi, H1, H2): xi = [dR dP dV], dP / dV in the BODY frame of the state. */
static void nav_retract(const fs_nav *X, const double *xi, fs_nav *Y, double *H1 /*9x9*/, double *H2 /*9x9*/) {
    double bRc[9], D_bRc_xi[9], nRc[9], bRcT[9], nRcT[9], tmp[3], S[9], D[9];
    so3_expmap(xi, bRc, D_bRc_xi);
    f_mm(X->R, bRc, nRc);
    f_tr(bRc, bRcT);
    f_tr(nRc, nRcT);
    if (H1) {
        memset(H1, 0, 81 * sizeof(double));
        set9(H1, 9, 0, 0, bRcT, 1.0);                       /* D_R_nRb: Rot3::compose H1 = bRc^T */
        f_skew(xi + 3, S); f_mm(X->R, S, D);                /* D_t_nRb = -nRb [dP]x  (Rot3::rotate H1) */
        for (int i = 0; i < 9; i++) D[i] = -D[i];
        f_mm(nRcT, D, D);
        set9(H1, 9, 3, 0, D, 1.0);
        set9(H1, 9, 3, 3, bRcT, 1.0);
        f_skew(xi + 6, S); f_mm(X->R, S, D);
        for (int i = 0; i < 9; i++) D[i] = -D[i];
        f_mm(nRcT, D, D);
        set9(H1, 9, 6, 0, D, 1.0);
        set9(H1, 9, 6, 6, bRcT, 1.0);
    }
    if (H2) {
        memset(H2, 0, 81 * sizeof(double));
        set9(H2, 9, 0, 0, D_bRc_xi, 1.0);
        set9(H2, 9, 3, 3, bRcT, 1.0);
        set9(H2, 9, 6, 6, bRcT, 1.0);
    }
    fs_nav out;
    memcpy(out.R, nRc, sizeof nRc);
    f_mv(X->R, xi + 3, tmp); for (int i = 0; i < 3; i++) out.t[i] = X->t[i] + tmp[i];
    f_mv(X->R, xi + 6, tmp); for (int i = 0; i < 3; i++) out.v[i] = X->v[i] + tmp[i];
    *Y = out;
}

/* gtsam::NavState::update(b_acceleration, b_omega, dt, F, G1, G2) */
static void nav_update(const fs_nav *X, const double *acc, const double *om, double dt, fs_nav *Y,
                       double *A /*9x9*/, double *B /*9x3*/, double *C /*9x3*/) {
    double RT[9], b_v[3], D_xiP_state[27], xi[9], H1[81], H2[81], S[9];
    const double dt22 = 0.5 * dt * dt;
    f_tr(X->R, RT);
    f_mv(RT, X->v, b_v);                                    /* bodyVelocity, H = [skew(b_v) 0 I] */
    memset(D_xiP_state, 0, sizeof D_xiP_state);
    f_skew(b_v, S);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) D_xiP_state[i * 9 + j] = S[i * 3 + j];
    for (int i = 0; i < 3; i++) D_xiP_state[i * 9 + 6 + i] = 1.0;
    for (int i = 0; i < 3; i++) {
        xi[i] = dt * om[i];
        xi[3 + i] = dt * b_v[i] + dt22 * acc[i];
        xi[6 + i] = dt * acc[i];
    }
    nav_retract(X, xi, Y, H1, H2);
    if (A) {
        memcpy(A, H1, sizeof H1);
        /* F.middleRows<3>(3) += dt * D_newState_xi(3:6,3:6) * D_xiP_state */
        for (int i = 0; i < 3; i++) for (int j = 0; j < 9; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += H2[(3 + i) * 9 + 3 + k] * D_xiP_state[k * 9 + j];
            A[(3 + i) * 9 + j] += dt * s;
        }
    }
    if (B) for (int i = 0; i < 9; i++) for (int j = 0; j < 3; j++)
        B[i * 3 + j] = H2[i * 9 + 3 + j] * dt22 + H2[i * 9 + 6 + j] * dt;
    if (C) for (int i = 0; i < 9; i++) for (int j = 0; j < 3; j++) C[i * 3 + j] = H2[i * 9 + j] * dt;
}

typedef struct {
    fs_nav X;
    double dT;
    double delRdelBiasOmega[9], delPdelBiasAcc[9], delPdelBiasOmega[9], delVdelBiasAcc[9], delVdelBiasOmega[9];
    double P[225];   /* preintMeasCov, order [R p v b_a b_g] */
} fs_state;

/* PreintegratedCombinedMeasurements::integrateMeasurement (ManifoldPreintegration::update + covariance) */
static void fs_integrate(fs_state *s, const cpi_oracle_params *prm, const double *bg, const double *ba,
                         const double *meas_acc, const double *meas_om, double dt) {
    double acc[3], om[3], oldR[9], A[81], B[27], C[27];
    for (int i = 0; i < 3; i++) { acc[i] = meas_acc[i] - ba[i]; om[i] = meas_om[i] - bg[i]; }
    memcpy(oldR, s->X.R, sizeof oldR);
    s->dT += dt;
    fs_nav Y;
    nav_update(&s->X, acc, om, dt, &Y, A, B, C);
    s->X = Y;

    /* bias Jacobians (ManifoldPreintegration::update) */
    double S[9], D_acc_R[9], D_acc_biasOmega[9], iom[3], incrR[9], D_incrR[9], incrRt[9], T[9];
    f_skew(acc, S); f_mm(oldR, S, D_acc_R);
    for (int i = 0; i < 9; i++) D_acc_R[i] = -D_acc_R[i];
    f_mm(D_acc_R, s->delRdelBiasOmega, D_acc_biasOmega);
    for (int i = 0; i < 3; i++) iom[i] = om[i] * dt;
    so3_expmap(iom, incrR, D_incrR);
    f_tr(incrR, incrRt);
    f_mm(incrRt, s->delRdelBiasOmega, T);
    for (int i = 0; i < 9; i++) s->delRdelBiasOmega[i] = T[i] - D_incrR[i] * dt;
    const double dt22 = 0.5 * dt * dt;
    for (int i = 0; i < 9; i++) {
        s->delPdelBiasAcc[i] += s->delVdelBiasAcc[i] * dt - dt22 * oldR[i];
        s->delPdelBiasOmega[i] += dt * s->delVdelBiasOmega[i] + dt22 * D_acc_biasOmega[i];
        s->delVdelBiasAcc[i] += -oldR[i] * dt;
        s->delVdelBiasOmega[i] += D_acc_biasOmega[i] * dt;
    }

    /* covariance: F (15x15), G_measCov_Gt, P = F P F^T + G   (GTSAM order R0 p3 v6 a9 g12) */
    double F[225], G[225], FP[225], th_H[9], pos_H[9], vel_H[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        th_H[i * 3 + j] = -C[i * 3 + j];
        pos_H[i * 3 + j] = -B[(3 + i) * 3 + j];
        vel_H[i * 3 + j] = -B[(6 + i) * 3 + j];
    }
    memset(F, 0, sizeof F);
    for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) F[i * 15 + j] = A[i * 9 + j];
    set9(F, 15, 0, 12, th_H, 1.0);
    /* F(p, b_a) stays ZERO.  CombinedImuFactor.cpp of the GTSAM the reference pins (commit c21186c6, GTSAM 4.0 era) sets
     * only F.block<3,3>(0,12) = theta_H_biasOmega and F.block<3,3>(6,9) = vel_H_biasAcc and carries the comment
     * "TODO(frank): should we not also account for bias on position?"; pos_H_biasAcc = -B.middleRows<3>(3) entered GTSAM
     * years later.  (Restated from the publication history -- GTSAM is not in the tree; round-1 had the block, the
     * round-2 advisor flagged it.  With the block P(p, b_a) and what couples to it differ by ~1/n.) */
    (void)pos_H;
    set9(F, 15, 6, 9, vel_H, 1.0);
    for (int i = 9; i < 15; i++) F[i * 15 + i] = 1.0;
    memset(G, 0, sizeof G);
    const double wCov = prm->sigma_w * prm->sigma_w, aCov = prm->sigma_a * prm->sigma_a;
    const double bgCov = prm->sigma_wb * prm->sigma_wb, baCov = prm->sigma_ab * prm->sigma_ab;
    {   /* D_v_v = (1/dt) vel_H aCov vel_H^T ; D_R_R = (1/dt) th_H wCov th_H^T ; D_t_t = dt * integrationCovariance (= 0,
         * GraphSolver_IMU.cpp:158) ; biasAccOmegaInt = 0 (:159) */
        double Tt[9], Q[9];
        f_tr(vel_H, Tt);
        for (int i = 0; i < 9; i++) Q[i] = vel_H[i] * aCov;
        f_mm(Q, Tt, Q); set9(G, 15, 6, 6, Q, 1.0 / dt);
        f_tr(th_H, Tt);
        for (int i = 0; i < 9; i++) Q[i] = th_H[i] * wCov;
        f_mm(Q, Tt, Q); set9(G, 15, 0, 0, Q, 1.0 / dt);
        for (int i = 0; i < 3; i++) { G[(9 + i) * 15 + 9 + i] = dt * baCov; G[(12 + i) * 15 + 12 + i] = dt * bgCov; }
    }
    for (int i = 0; i < 15; i++) for (int j = 0; j < 15; j++) {
        double x = 0;
        for (int k = 0; k < 15; k++) x += F[i * 15 + k] * s->P[k * 15 + j];
        FP[i * 15 + j] = x;
    }
    for (int i = 0; i < 15; i++) for (int j = 0; j < 15; j++) {
        double x = 0;
        for (int k = 0; k < 15; k++) x += FP[i * 15 + k] * F[j * 15 + k];
        s->P[i * 15 + j] = x + G[i * 15 + j];
    }
}

/* One window, then the call-site conversions of GraphSolver_IMU.cpp:201-231. */
void cpi_oracle_forster_window(const cpi_oracle_params *prm, int n, const double *knots, const double *lin,
                               cpi_oracle_out *out) {
    fs_state s;
    memset(&s, 0, sizeof s);
    f_eye(s.X.R);
    const double *bg = lin, *ba = lin + 3;
    for (int i = 0; i < n; i++) {
        const double *k0 = knots + 7 * i, *k1 = knots + 7 * (i + 1);
        const double dt = k1[0] - k0[0];
        if (dt > 0) fs_integrate(&s, prm, bg, ba, k0 + 4, k0 + 1, dt);   /* reading i held over [t_i, t_i+1] (:171-180) */
    }
    memset(out, 0, sizeof *out);
    out->DT = s.dT;
    for (int i = 0; i < 3; i++) { out->alpha[i] = s.X.t[i]; out->beta[i] = s.X.v[i]; }   /* deltaPij, deltaVij (:204-205) */
    double kplus_R_k[9];
    f_tr(s.X.R, kplus_R_k);                                                               /* :206 */
    cpi_oracle_rot_2_quat_rm(kplus_R_k, out->q);                                          /* :229 */
    /* column-major 3x3 outputs = transposes of the row-major internals */
    f_tr(kplus_R_k, out->R);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        out->J_q[j * 3 + i] = -s.delRdelBiasOmega[i * 3 + j];                             /* :210 */
        out->H_a[j * 3 + i] = s.delPdelBiasAcc[i * 3 + j];                                /* :211 */
        out->J_a[j * 3 + i] = s.delPdelBiasOmega[i * 3 + j];                              /* :212 */
        out->H_b[j * 3 + i] = s.delVdelBiasAcc[i * 3 + j];                                /* :213 */
        out->J_b[j * 3 + i] = s.delVdelBiasOmega[i * 3 + j];                              /* :214 */
    }
    /* swapcovariance(P, 1, 4) (:225, :240-254): block order [R p v ba bg] -> [R bg v ba p] */
    static const int perm[5] = { 0, 4, 2, 3, 1 };
    for (int bi = 0; bi < 5; bi++) for (int bj = 0; bj < 5; bj++)
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
            out->P[(bj * 3 + j) * 15 + bi * 3 + i] = s.P[(perm[bi] * 3 + i) * 15 + perm[bj] * 3 + j];
}

/* ---- test hooks (finite-difference pinning of A, B, C): state = {R row-major 9, t 3, v 3} */
void cpi_oracle_navstate_retract(const double *state15, const double *xi9, double *out15) {
    fs_nav X, Y;
    memcpy(X.R, state15, 72); memcpy(X.t, state15 + 9, 24); memcpy(X.v, state15 + 12, 24);
    nav_retract(&X, xi9, &Y, NULL, NULL);
    memcpy(out15, Y.R, 72); memcpy(out15 + 9, Y.t, 24); memcpy(out15 + 12, Y.v, 24);
}
void cpi_oracle_navstate_update(const double *state15, const double *acc, const double *om, double dt,
                                double *out15, double *A81, double *B27, double *C27) {
    fs_nav X, Y;
    memcpy(X.R, state15, 72); memcpy(X.t, state15 + 9, 24); memcpy(X.v, state15 + 12, 24);
    nav_update(&X, acc, om, dt, &Y, A81, B27, C27);
    memcpy(out15, Y.R, 72); memcpy(out15 + 9, Y.t, 24); memcpy(out15 + 12, Y.v, 24);
}
