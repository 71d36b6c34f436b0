/*
 * cpi_oracle.c -- CPU restatement (plain C99) of the rpng/cpi preintegration hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see cpi_oracle.h).  Follows the reference operation order:
 *   quat_ops.h:45-197, CpiBase.h:52-124, CpiV1.h:62-361, CpiV2.h:84-467,
 *   ImuFactorCPIv1.cpp:37-208, ImuFactorCPIv2.cpp:38-212, JPLNavState.cpp:37-88,
 *   GraphSolver_IMU.cpp:50-69,263-307.
 * The covariance is propagated "as written": dense 15x15 / 21x21 products, classic RK4.
 * Internally matrices are row-major; the public API converts to column-major.
 */
#include "cpi_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- 3x3 helpers (row-major) */
static void eye3(double *A) { memset(A, 0, 9 * sizeof(double)); A[0] = A[4] = A[8] = 1.0; }
static void zero3(double *A) { memset(A, 0, 9 * sizeof(double)); }
/* quat_ops.h:92-98 */
static void skew_x(const double *w, double *S) {
    S[0] = 0;     S[1] = -w[2]; S[2] = w[1];
    S[3] = w[2];  S[4] = 0;     S[5] = -w[0];
    S[6] = -w[1]; S[7] = w[0];  S[8] = 0;
}
static void mm3(const double *A, const double *B, double *C) {
    double T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
            T[i * 3 + j] = s;
        }
    memcpy(C, T, sizeof T);
}
static void mt3(const double *A, double *At) {
    double T[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T[j * 3 + i] = A[i * 3 + j];
    memcpy(At, T, sizeof T);
}
static void mv3(const double *A, const double *x, double *y) {
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = A[i * 3] * x[0] + A[i * 3 + 1] * x[1] + A[i * 3 + 2] * x[2];
    y[0] = t[0]; y[1] = t[1]; y[2] = t[2];
}
static double norm3(const double *x) { return sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]); }
/* C = a*A + b*B + c*Cc (element-wise) */
static void lin3(double a, const double *A, double b, const double *B, double c, const double *Cc, double *out) {
    for (int i = 0; i < 9; i++) out[i] = a * A[i] + b * B[i] + c * Cc[i];
}
static void to_colmajor3(const double *A, double *out) { mt3(A, out); }

/* quat_ops.h:45-86 */
static void rot_2_quat(const double *rot, double *q) {
    double T = rot[0] + rot[4] + rot[8];
    double r00 = rot[0], r11 = rot[4], r22 = rot[8];
    if ((r00 >= T) && (r00 >= r11) && (r00 >= r22)) {
        q[0] = sqrt((1 + (2 * r00) - T) / 4);
        q[1] = (1 / (4 * q[0])) * (rot[1] + rot[3]);
        q[2] = (1 / (4 * q[0])) * (rot[2] + rot[6]);
        q[3] = (1 / (4 * q[0])) * (rot[5] - rot[7]);
    } else if ((r11 >= T) && (r11 >= r00) && (r11 >= r22)) {
        q[1] = sqrt((1 + (2 * r11) - T) / 4);
        q[0] = (1 / (4 * q[1])) * (rot[1] + rot[3]);
        q[2] = (1 / (4 * q[1])) * (rot[5] + rot[7]);
        q[3] = (1 / (4 * q[1])) * (rot[6] - rot[2]);
    } else if ((r22 >= T) && (r22 >= r00) && (r22 >= r11)) {
        q[2] = sqrt((1 + (2 * r22) - T) / 4);
        q[0] = (1 / (4 * q[2])) * (rot[2] + rot[6]);
        q[1] = (1 / (4 * q[2])) * (rot[5] + rot[7]);
        q[3] = (1 / (4 * q[2])) * (rot[1] - rot[3]);
    } else {
        q[3] = sqrt((1 + T) / 4);
        q[0] = (1 / (4 * q[3])) * (rot[5] - rot[7]);
        q[1] = (1 / (4 * q[3])) * (rot[6] - rot[2]);
        q[2] = (1 / (4 * q[3])) * (rot[1] - rot[3]);
    }
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

/* quat_ops.h:104-109 */
static void quat_2_Rot(const double *q, double *R) {
    double qx[9];
    skew_x(q, qx);
    double c = 2 * pow(q[3], 2) - 1;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[i * 3 + j] = c * (i == j ? 1.0 : 0.0) - 2 * q[3] * qx[i * 3 + j] + 2 * q[i] * q[j];
}

/* quat_ops.h:115-128 */
static void quat_multiply(const double *q, const double *p, double *out) {
    double Qm[16], qx[9], t[4];
    skew_x(q, qx);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Qm[i * 4 + j] = q[3] * (i == j ? 1.0 : 0.0) - qx[i * 3 + j];
        Qm[i * 4 + 3] = q[i];
        Qm[3 * 4 + i] = -q[i];
    }
    Qm[15] = q[3];
    for (int i = 0; i < 4; i++) {
        double s = 0;
        for (int k = 0; k < 4; k++) s += Qm[i * 4 + k] * p[k];
        t[i] = s;
    }
    if (t[3] < 0) { t[0] *= -1; t[1] *= -1; t[2] *= -1; t[3] *= -1; }
    double n = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3]);
    for (int i = 0; i < 4; i++) out[i] = t[i] / n;
}

/* quat_ops.h:145-162 */
static void Exp_so3(const double *w, double *R) {
    double wx[9], wx2[9], I[9];
    skew_x(w, wx);
    double theta = norm3(w);
    eye3(I);
    if (theta == 0) { eye3(R); return; }
    mm3(wx, wx, wx2);
    lin3(1.0, I, sin(theta) / theta, wx, (1 - cos(theta)) / pow(theta, 2), wx2, R);
}

/* quat_ops.h:190-197 */
static void quat_inv(const double *q, double *qi) { qi[0] = -q[0]; qi[1] = -q[1]; qi[2] = -q[2]; qi[3] = q[3]; }

/* ---------------------------------------------------------------- dense n x n helpers */
static void dmm(int n, const double *A, const double *B, double *C) { /* C = A*B */
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int k = 0; k < n; k++) s += A[i * n + k] * B[k * n + j];
            C[i * n + j] = s;
        }
}
static void dmmt(int n, const double *A, const double *B, double *C) { /* C = A*B^T */
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int k = 0; k < n; k++) s += A[i * n + k] * B[j * n + k];
            C[i * n + j] = s;
        }
}
static void set_block(int n, double *M, int r, int c, const double *B3, double scale) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[(r + i) * n + c + j] = scale * B3[i * 3 + j];
}
static void get_block(int n, const double *M, int r, int c, double *B3) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) B3[i * 3 + j] = M[(r + i) * n + c + j];
}

/* ---------------------------------------------------------------- preintegrator state */
typedef struct {
    cpi_oracle_params prm;
    double b_w[3], b_a[3], q_k_lin[4];
    double DT, alpha[3], beta[3], q[4], R[9];
    double J_q[9], J_a[9], J_b[9], H_a[9], H_b[9], O_a[9], O_b[9];
    double Qc[4];        /* sigma^2 of the four 3x3 diagonal blocks of Q_c (CpiBase.h:54-57) */
    double P[225];       /* P_meas, row-major */
    double Pbig[441];    /* CpiV2.h:46 */
    double D[441];       /* Discrete_J_b, CpiV2.h:49 */
} cpi_state;

static void state_init(cpi_state *s, const cpi_oracle_params *prm, const double *lin, const double *q_k_lin) {
    memset(s, 0, sizeof *s);
    s->prm = *prm;
    s->Qc[0] = pow(prm->sigma_w, 2);
    s->Qc[1] = pow(prm->sigma_wb, 2);
    s->Qc[2] = pow(prm->sigma_a, 2);
    s->Qc[3] = pow(prm->sigma_ab, 2);
    for (int i = 0; i < 3; i++) { s->b_w[i] = lin[i]; s->b_a[i] = lin[3 + i]; }
    if (q_k_lin) memcpy(s->q_k_lin, q_k_lin, 4 * sizeof(double));
    eye3(s->R);
    for (int i = 0; i < 21; i++) s->D[i * 21 + i] = 1.0;
    /* q_k2tau is uninitialised in the reference until the first non-degenerate feed_IMU
     * (CpiBase.h:102); we start it at identity, the value rot_2_quat(I) would give. */
    s->q[3] = 1.0;
}

/* G*Qc*G^T for the block structure of CpiV1.h:283-287 / CpiV2.h:340-344: G(0,0)=-I, G(3,3)=I,
 * G(6,6)=-Rs^T, G(9,9)=I, so the product is blkdiag(s_w^2 I, s_wb^2 I, s_a^2 Rs^T Rs, s_ab^2 I, 0).
 * Computed with the explicit Rs^T*Rs product to mirror the reference's rounding. */
static void add_GQGt(int n, const double *Qc, const double *Rs, double *Pdot) {
    double RsT[9], RtR[9];
    mt3(Rs, RsT);
    mm3(RsT, Rs, RtR);
    for (int i = 0; i < 3; i++) {
        Pdot[(0 + i) * n + 0 + i] += Qc[0];
        Pdot[(3 + i) * n + 3 + i] += Qc[1];
        Pdot[(9 + i) * n + 9 + i] += Qc[3];
        for (int j = 0; j < 3; j++) Pdot[(6 + i) * n + 6 + j] += Qc[2] * RtR[i * 3 + j];
    }
}

/* Pdot = F*P + P*F^T + G Q G^T */
static void lyap_rhs(int n, const double *F, const double *P, const double *Qc, const double *Rs, double *Pdot, double *tmp) {
    dmm(n, F, P, Pdot);
    dmmt(n, P, F, tmp);
    for (int i = 0; i < n * n; i++) Pdot[i] += tmp[i];
    add_GQGt(n, Qc, Rs, Pdot);
}

/* State Jacobian of CpiV1.h:276-281 (n=15) / CpiV2.h:331-338 (n=21). */
static void build_F(int n, const double *w_x, const double *a_x, const double *Rs, const double *R_k2tau,
                    const double *R_G_to_k, const double *grav, double *F) {
    double RsT[9], T[9], I[9];
    memset(F, 0, (size_t)n * n * sizeof(double));
    eye3(I);
    mt3(Rs, RsT);
    set_block(n, F, 0, 0, w_x, -1.0);
    set_block(n, F, 0, 3, I, -1.0);
    mm3(RsT, a_x, T);
    set_block(n, F, 6, 0, T, -1.0);
    set_block(n, F, 6, 9, RsT, -1.0);
    set_block(n, F, 12, 6, I, 1.0);
    if (n == 21) {
        double Rg[3], g_k[3], g_tau[3], S[9], T2[9];
        mv3(R_G_to_k, grav, g_k);          /* R_G_to_k * grav */
        {   /* R_k2tau * R_G_to_k * grav : Eigen evaluates (R_k2tau*R_G_to_k)*grav left-to-right */
            double RR[9];
            mm3(R_k2tau, R_G_to_k, RR);
            mv3(RR, grav, g_tau);
        }
        (void)Rg;
        skew_x(g_tau, S);
        mm3(RsT, S, T);
        set_block(n, F, 6, 15, T, -1.0);
        skew_x(g_k, S);
        mm3(RsT, R_k2tau, T2);
        mm3(T2, S, T);
        set_block(n, F, 6, 18, T, -1.0);
    }
}

/* One feed_IMU call.  model 1: CpiV1.h:62-361; model 2: CpiV2.h:84-467. */
static void feed_imu(cpi_state *s, double t_0, double t_1, const double *w_m_0, const double *a_m_0,
                     const double *w_m_1, const double *a_m_1) {
    const int v2 = (s->prm.model == 2);
    const double *grav = s->prm.grav;
    double I[9];
    eye3(I);

    double delta_t = t_1 - t_0;
    s->DT += delta_t;
    if (delta_t == 0) return;

    double R_G_to_k[9];
    zero3(R_G_to_k);
    double w_hat[3], a_hat[3];
    for (int i = 0; i < 3; i++) { w_hat[i] = w_m_0[i] - s->b_w[i]; a_hat[i] = a_m_0[i] - s->b_a[i]; }
    if (v2) {
        /* CpiV2.h:99  a_hat = a_m_0 - b_a_lin - R_k2tau*quat_2_Rot(q_k_lin)*grav */
        double RR[9], g[3];
        quat_2_Rot(s->q_k_lin, R_G_to_k);
        mm3(s->R, R_G_to_k, RR);
        mv3(RR, grav, g);
        for (int i = 0; i < 3; i++) a_hat[i] = a_m_0[i] - s->b_a[i] - g[i];
    }
    if (s->prm.imu_avg) {
        for (int i = 0; i < 3; i++) { w_hat[i] += w_m_1[i] - s->b_w[i]; w_hat[i] = 0.5 * w_hat[i]; }
        if (!v2)
            for (int i = 0; i < 3; i++) { a_hat[i] += a_m_1[i] - s->b_a[i]; a_hat[i] = .5 * a_hat[i]; }
    }

    double w_hatdt[3] = { w_hat[0] * delta_t, w_hat[1] * delta_t, w_hat[2] * delta_t };
    double w_1 = w_hat[0], w_2 = w_hat[1], w_3 = w_hat[2];
    double mag_w = norm3(w_hat);
    double w_dt = mag_w * delta_t;
    int small_w = (mag_w < 0.008726646);
    double dt_2 = pow(delta_t, 2);
    double cos_wt = cos(w_dt);
    double sin_wt = sin(w_dt);

    double w_x[9], a_x[9], w_tx[9], w_x_2[9], w_tx_2[9];
    skew_x(w_hat, w_x);
    skew_x(w_hatdt, w_tx);
    mm3(w_x, w_x, w_x_2);
    mm3(w_tx, w_tx, w_tx_2);

    /* ---- measurement means ---- */
    double R_tau2tau1[9];
    if (small_w) lin3(1.0, I, -delta_t, w_x, (pow(delta_t, 2) / 2), w_x_2, R_tau2tau1);
    else lin3(1.0, I, -(sin_wt / mag_w), w_x, ((1.0 - cos_wt) / (pow(mag_w, 2.0))), w_x_2, R_tau2tau1);

    double R_k2tau1[9], R_tau12k[9];
    mm3(R_tau2tau1, s->R, R_k2tau1);
    mt3(R_k2tau1, R_tau12k);

    if (v2 && s->prm.imu_avg) {
        /* CpiV2.h:146-149: average the LOCAL acceleration */
        double RR[9], g[3];
        mm3(R_k2tau1, R_G_to_k, RR);
        mv3(RR, grav, g);
        for (int i = 0; i < 3; i++) { a_hat[i] += a_m_1[i] - s->b_a[i] - g[i]; a_hat[i] = 0.5 * a_hat[i]; }
    }
    skew_x(a_hat, a_x);

    double f_1, f_2, f_3, f_4;
    if (small_w) {
        f_1 = -(pow(delta_t, 3) / 3);
        f_2 = (pow(delta_t, 4) / 8);
        f_3 = -(pow(delta_t, 2) / 2);
        f_4 = (pow(delta_t, 3) / 6);
    } else {
        f_1 = (w_dt * cos_wt - sin_wt) / (pow(mag_w, 3));
        f_2 = (pow(w_dt, 2) - 2 * cos_wt - 2 * w_dt * sin_wt + 2) / (2 * pow(mag_w, 4));
        f_3 = -(1 - cos_wt) / pow(mag_w, 2);
        f_4 = (w_dt - sin_wt) / pow(mag_w, 3);
    }

    double alpha_arg[9], Beta_arg[9], H_al[9], H_be[9];
    lin3((dt_2 / 2.0), I, f_1, w_x, f_2, w_x_2, alpha_arg);
    lin3(delta_t, I, f_3, w_x, f_4, w_x_2, Beta_arg);
    mm3(R_tau12k, alpha_arg, H_al);
    mm3(R_tau12k, Beta_arg, H_be);

    {
        double ta[3], tb[3];
        mv3(H_al, a_hat, ta);
        mv3(H_be, a_hat, tb);
        for (int i = 0; i < 3; i++) s->alpha[i] += s->beta[i] * delta_t + ta[i]; /* old beta */
        for (int i = 0; i < 3; i++) s->beta[i] += tb[i];
    }

    /* ---- bias Jacobians (analytical) ---- */
    double J_r_tau1[9];
    if (small_w) lin3(1.0, I, -.5, w_tx, (1.0 / 6.0), w_tx_2, J_r_tau1);
    else lin3(1.0, I, -((1 - cos_wt) / (pow((w_dt), 2.0))), w_tx, ((w_dt - sin_wt) / (pow(w_dt, 3.0))), w_tx_2, J_r_tau1);

    double J_save[9];
    memcpy(J_save, s->J_q, sizeof J_save);
    {
        double T[9];
        mm3(R_tau2tau1, s->J_q, T);
        for (int i = 0; i < 9; i++) s->J_q[i] = T[i] + J_r_tau1[i] * delta_t;
    }
    for (int i = 0; i < 9; i++) s->H_a[i] -= H_al[i];
    for (int i = 0; i < 9; i++) s->H_a[i] += delta_t * s->H_b[i];
    for (int i = 0; i < 9; i++) s->H_b[i] -= H_be[i];

    double g_k[3] = {0, 0, 0}, g_tau[3] = {0, 0, 0};
    if (v2) {
        /* CpiV2.h:201-205 */
        double S[9], T[9], T2[9];
        mv3(R_G_to_k, grav, g_k);
        skew_x(g_k, S);
        for (int i = 0; i < 9; i++) s->O_a[i] += delta_t * s->O_b[i];
        mm3(H_al, s->R, T); mm3(T, S, T2);
        for (int i = 0; i < 9; i++) s->O_a[i] += -T2[i];
        mm3(H_be, s->R, T); mm3(T, S, T2);
        for (int i = 0; i < 9; i++) s->O_b[i] += -T2[i];
        /* CpiV2.h:274 g_tau = R_k2tau*quat_2_Rot(q_k_lin)*grav */
        double RR[9];
        mm3(s->R, R_G_to_k, RR);
        mv3(RR, grav, g_tau);
    }

    double df_dw[4];
    if (small_w) {
        df_dw[0] = -(pow(delta_t, 5) / 15);
        df_dw[1] = (pow(delta_t, 6) / 72);
        df_dw[2] = -(pow(delta_t, 4) / 12);
        df_dw[3] = (pow(delta_t, 5) / 60);
    } else {
        df_dw[0] = (pow(w_dt, 2) * sin_wt - 3 * sin_wt + 3 * w_dt * cos_wt) / pow(mag_w, 5);
        df_dw[1] = (pow(w_dt, 2) - 4 * cos_wt - 4 * w_dt * sin_wt + pow(w_dt, 2) * cos_wt + 4) / (pow(mag_w, 6));
        df_dw[2] = (2 * (cos_wt - 1) + w_dt * sin_wt) / (pow(mag_w, 4));
        df_dw[3] = (2 * w_dt + w_dt * cos_wt - 3 * sin_wt) / (pow(mag_w, 5));
    }
    const double wv[3] = { w_1, w_2, w_3 };

    for (int i = 0; i < 9; i++) s->J_a[i] += s->J_b[i] * delta_t; /* old J_b */
    for (int c = 0; c < 3; c++) {
        double e[3] = {0, 0, 0}, ex[9], Jqe[3], Sk[9], dR[9];
        e[c] = 1.0;
        skew_x(e, ex);
        mv3(s->J_q, e, Jqe); /* new J_q */
        skew_x(Jqe, Sk);
        mm3(R_tau12k, Sk, dR);
        for (int i = 0; i < 9; i++) dR[i] = -dR[i];
        double exwx[9], wxex[9], sym[9];
        mm3(ex, w_x, exwx);
        mm3(w_x, ex, wxex);
        for (int i = 0; i < 9; i++) sym[i] = exwx[i] + wxex[i];
        double df1 = wv[c] * df_dw[0], df2 = wv[c] * df_dw[1], df3 = wv[c] * df_dw[2], df4 = wv[c] * df_dw[3];
        double Ga[9], Gb[9], Ta[9], Tb[9], Ma[9], Mb[9], ca[3], cb[3];
        for (int i = 0; i < 9; i++) {
            Ga[i] = df1 * w_x[i] - f_1 * ex[i] + df2 * w_x_2[i] - f_2 * sym[i];
            Gb[i] = df3 * w_x[i] - f_3 * ex[i] + df4 * w_x_2[i] - f_4 * sym[i];
        }
        mm3(dR, alpha_arg, Ta); mm3(R_tau12k, Ga, Ma);
        mm3(dR, Beta_arg, Tb);  mm3(R_tau12k, Gb, Mb);
        for (int i = 0; i < 9; i++) { Ma[i] += Ta[i]; Mb[i] += Tb[i]; }
        mv3(Ma, a_hat, ca);
        mv3(Mb, a_hat, cb);
        if (v2) {
            /* CpiV2.h:282-305: extra -H_al*skew_x(J_save*e_i)*g_tau terms; the J_b column 0
             * term carries a double minus in the reference (:296-297) and is reproduced. */
            double Jse[3], Ss[9], T[9], ga[3], gb[3];
            mv3(J_save, e, Jse);
            skew_x(Jse, Ss);
            mm3(H_al, Ss, T); mv3(T, g_tau, ga);
            mm3(H_be, Ss, T); mv3(T, g_tau, gb);
            for (int i = 0; i < 3; i++) ca[i] -= ga[i];
            if (c == 0) for (int i = 0; i < 3; i++) cb[i] = cb[i] - (-gb[i]);
            else        for (int i = 0; i < 3; i++) cb[i] -= gb[i];
        }
        for (int i = 0; i < 3; i++) { s->J_a[i * 3 + c] += ca[i]; s->J_b[i * 3 + c] += cb[i]; }
    }

    /* ---- measurement covariance (RK4) ---- */
    double R_mid[9];
    {
        double Rm[9];
        if (!v2) {
            if (small_w) lin3(1.0, I, -.5 * delta_t, w_x, (pow(.5 * delta_t, 2) / 2), w_x_2, Rm);
            else lin3(1.0, I, -(sin(mag_w * .5 * delta_t) / mag_w), w_x,
                      ((1.0 - cos(mag_w * .5 * delta_t)) / (pow(mag_w, 2.0))), w_x_2, Rm);
        } else {
            double dt_mid = delta_t / 2.0;
            double w_dt_mid = mag_w * dt_mid;
            if (small_w) lin3(1.0, I, -dt_mid, w_x, (pow(dt_mid, 2) / 2), w_x_2, Rm);
            else lin3(1.0, I, -(sin(w_dt_mid) / mag_w), w_x, ((1.0 - cos(w_dt_mid)) / (pow(mag_w, 2.0))), w_x_2, Rm);
        }
        mm3(Rm, s->R, R_mid);
    }

    if (!v2) {
        enum { n = 15, nn = 225 };
        double F1[nn], F2[nn], F4[nn], k1[nn], k2[nn], k3[nn], k4[nn], Pk[nn], tmp[nn];
        build_F(n, w_x, a_x, s->R, s->R, NULL, NULL, F1);
        build_F(n, w_x, a_x, R_mid, s->R, NULL, NULL, F2);
        build_F(n, w_x, a_x, R_k2tau1, s->R, NULL, NULL, F4);
        lyap_rhs(n, F1, s->P, s->Qc, s->R, k1, tmp);
        for (int i = 0; i < nn; i++) Pk[i] = s->P[i] + k1[i] * delta_t / 2.0;
        lyap_rhs(n, F2, Pk, s->Qc, R_mid, k2, tmp);
        for (int i = 0; i < nn; i++) Pk[i] = s->P[i] + k2[i] * delta_t / 2.0;
        lyap_rhs(n, F2, Pk, s->Qc, R_mid, k3, tmp);
        for (int i = 0; i < nn; i++) Pk[i] = s->P[i] + k3[i] * delta_t;
        lyap_rhs(n, F4, Pk, s->Qc, R_k2tau1, k4, tmp);
        for (int i = 0; i < nn; i++) s->P[i] += (delta_t / 6.0) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) {
                double m = 0.5 * (s->P[i * n + j] + s->P[j * n + i]);
                s->P[i * n + j] = m; s->P[j * n + i] = m;
            }
        for (int i = 0; i < n; i++) s->P[i * n + i] = 0.5 * (s->P[i * n + i] + s->P[i * n + i]);
    } else {
        enum { n = 21, nn = 441 };
        double dt_mid = delta_t / 2.0;
        double F1[nn], F2[nn], F4[nn], k1[nn], k2[nn], k3[nn], k4[nn], Pk[nn], tmp[nn];
        double Ph1[nn], Ph2[nn], Ph3[nn], Ph4[nn], Phk[nn], Phi[nn];
        build_F(n, w_x, a_x, s->R, s->R, R_G_to_k, grav, F1);
        build_F(n, w_x, a_x, R_mid, s->R, R_G_to_k, grav, F2);
        build_F(n, w_x, a_x, R_k2tau1, s->R, R_G_to_k, grav, F4);
        /* k1 */
        memcpy(Ph1, F1, sizeof Ph1);
        lyap_rhs(n, F1, s->Pbig, s->Qc, s->R, k1, tmp);
        /* k2 */
        for (int i = 0; i < nn; i++) Phk[i] = Ph1[i] * dt_mid;
        for (int i = 0; i < n; i++) Phk[i * n + i] = 1.0 + Ph1[i * n + i] * dt_mid;
        for (int i = 0; i < nn; i++) Pk[i] = s->Pbig[i] + k1[i] * dt_mid;
        dmm(n, F2, Phk, Ph2);
        lyap_rhs(n, F2, Pk, s->Qc, R_mid, k2, tmp);
        /* k3 */
        for (int i = 0; i < nn; i++) Phk[i] = Ph2[i] * dt_mid;
        for (int i = 0; i < n; i++) Phk[i * n + i] = 1.0 + Ph2[i * n + i] * dt_mid;
        for (int i = 0; i < nn; i++) Pk[i] = s->Pbig[i] + k2[i] * dt_mid;
        dmm(n, F2, Phk, Ph3);
        lyap_rhs(n, F2, Pk, s->Qc, R_mid, k3, tmp);
        /* k4 */
        for (int i = 0; i < nn; i++) Phk[i] = Ph3[i] * delta_t;
        for (int i = 0; i < n; i++) Phk[i * n + i] = 1.0 + Ph3[i * n + i] * delta_t;
        for (int i = 0; i < nn; i++) Pk[i] = s->Pbig[i] + k3[i] * delta_t;
        dmm(n, F4, Phk, Ph4);
        lyap_rhs(n, F4, Pk, s->Qc, R_k2tau1, k4, tmp);
        /* collect */
        for (int i = 0; i < nn; i++) s->Pbig[i] += (delta_t / 6.0) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) {
                double m = 0.5 * (s->Pbig[i * n + j] + s->Pbig[j * n + i]);
                s->Pbig[i * n + j] = m; s->Pbig[j * n + i] = m;
            }
        for (int i = 0; i < nn; i++) Phi[i] = (delta_t / 6.0) * (Ph1[i] + 2.0 * Ph2[i] + 2.0 * Ph3[i] + Ph4[i]);
        for (int i = 0; i < n; i++) Phi[i * n + i] = 1.0 + Phi[i * n + i];

        /* clone to new sample time / marginalise old (CpiV2.h:436-443) */
        double Bk[nn];
        memset(Bk, 0, sizeof Bk);
        for (int i = 0; i < n; i++) Bk[i * n + i] = 1.0;
        for (int i = 0; i < 3; i++) { Bk[(15 + i) * n + 15 + i] = 0.0; Bk[(15 + i) * n + i] = 1.0; }
        dmm(n, Bk, s->Pbig, tmp);
        dmmt(n, tmp, Bk, s->Pbig);
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) {
                double m = 0.5 * (s->Pbig[i * n + j] + s->Pbig[j * n + i]);
                s->Pbig[i * n + j] = m; s->Pbig[j * n + i] = m;
            }
        dmm(n, Bk, Phi, tmp);
        dmm(n, tmp, s->D, Phk);
        memcpy(s->D, Phk, sizeof Phk);
        for (int i = 0; i < 15; i++) for (int j = 0; j < 15; j++) s->P[i * 15 + j] = s->Pbig[i * n + j];

        if (s->prm.state_transition_jacobians) {
            get_block(n, s->D, 0, 3, s->J_q);
            for (int i = 0; i < 9; i++) s->J_q[i] = -s->J_q[i];
            get_block(n, s->D, 12, 3, s->J_a);
            get_block(n, s->D, 6, 3, s->J_b);
            get_block(n, s->D, 12, 9, s->H_a);
            get_block(n, s->D, 6, 9, s->H_b);
            get_block(n, s->D, 12, 18, s->O_a);
            get_block(n, s->D, 6, 18, s->O_b);
        }
    }

    memcpy(s->R, R_k2tau1, sizeof s->R);
    rot_2_quat(s->R, s->q);
}

static void state_export(const cpi_state *s, cpi_oracle_out *o) {
    o->DT = s->DT;
    for (int i = 0; i < 3; i++) { o->alpha[i] = s->alpha[i]; o->beta[i] = s->beta[i]; }
    for (int i = 0; i < 4; i++) o->q[i] = s->q[i];
    to_colmajor3(s->R, o->R);
    to_colmajor3(s->J_q, o->J_q); to_colmajor3(s->J_a, o->J_a); to_colmajor3(s->J_b, o->J_b);
    to_colmajor3(s->H_a, o->H_a); to_colmajor3(s->H_b, o->H_b);
    to_colmajor3(s->O_a, o->O_a); to_colmajor3(s->O_b, o->O_b);
    for (int i = 0; i < 15; i++) for (int j = 0; j < 15; j++) o->P[j * 15 + i] = s->P[i * 15 + j];
}

void cpi_oracle_rot_2_quat_rm(const double *rot_rowmajor, double *q) { rot_2_quat(rot_rowmajor, q); }

/* Test hook: the restated quat_ops.h helpers, same interface as cpi_ref_quat_ops (ref_shim.cpp); matrices ROW-major. */
int cpi_oracle_quat_ops(int op, long n, const double *in, double *out) {
    for (long k = 0; k < n; k++) {
        switch (op) {
            case 0: rot_2_quat(in + 9 * k, out + 4 * k); break;
            case 1: skew_x(in + 3 * k, out + 9 * k); break;
            case 2: quat_2_Rot(in + 4 * k, out + 9 * k); break;
            case 3: quat_multiply(in + 8 * k, in + 8 * k + 4, out + 4 * k); break;
            case 4: Exp_so3(in + 3 * k, out + 9 * k); break;
            case 5: quat_inv(in + 4 * k, out + 4 * k); break;
            default: return 1;
        }
    }
    return 0;
}

static void run_window(const cpi_oracle_params *prm, int n, const double *knots, const double *lin,
                       const double *q_k_lin, cpi_oracle_out *out, cpi_oracle_out *trace) {
    if (prm->model == 3) {   /* Forster comparator (forster_oracle.c); trace = the prefix windows */
        if (trace) for (int i = 0; i < n; i++) cpi_oracle_forster_window(prm, i + 1, knots, lin, trace + i);
        if (out) cpi_oracle_forster_window(prm, n, knots, lin, out);
        return;
    }
    cpi_state *s = (cpi_state *)malloc(sizeof(cpi_state));
    state_init(s, prm, lin, q_k_lin);
    for (int i = 0; i < n; i++) {
        const double *k0 = knots + 7 * i, *k1 = knots + 7 * (i + 1);
        double dt = k1[0] - k0[0];
        if (dt >= 0) feed_imu(s, k0[0], k1[0], k0 + 1, k0 + 4, k1 + 1, k1 + 4);
        if (trace) state_export(s, trace + i);
    }
    if (out) state_export(s, out);
    free(s);
}

void cpi_oracle_window(const cpi_oracle_params *prm, int n, const double *knots, const double *lin,
                       const double *q_k_lin, cpi_oracle_out *out) {
    run_window(prm, n, knots, lin, q_k_lin, out, NULL);
}

void cpi_oracle_window_trace(const cpi_oracle_params *prm, int n, const double *knots, const double *lin,
                             const double *q_k_lin, cpi_oracle_out *trace) {
    run_window(prm, n, knots, lin, q_k_lin, NULL, trace);
}

void cpi_oracle_batch(const cpi_oracle_params *prm, long W, int n, const double *knots, const double *lin,
                      const double *q_k_lin, cpi_oracle_out *out) {
    for (long w = 0; w < W; w++)
        run_window(prm, n, knots + (size_t)w * (n + 1) * 7, lin + (size_t)w * 6,
                   q_k_lin ? q_k_lin + (size_t)w * 4 : NULL, out + w, NULL);
}

/* GraphSolver_IMU.cpp:50-69 with the three deques replaced by a front index into the stream plus the
 * (possibly overwritten) front timestamp. */
void cpi_oracle_stream(const cpi_oracle_params *prm, long K, const double *stream, long U,
                       const double *update_times, const double *lin, const double *q_k_lin, cpi_oracle_out *out) {
    long front = 0;                       /* index of imu_*.at(0) in the stream */
    double front_t = stream[0];           /* imu_times.at(0): overwritten by updatetime after a tail interval */
    cpi_state *s = (cpi_state *)malloc(sizeof(cpi_state));
    for (long u = 0; u < U; u++) {
        const double updatetime = update_times[u];
        state_init(s, prm, lin + u * 6, q_k_lin ? q_k_lin + u * 4 : NULL);
        while ((K - front) > 1 && stream[(front + 1) * 7] <= updatetime) {
            const double *k0 = stream + front * 7, *k1 = stream + (front + 1) * 7;
            double dt = k1[0] - front_t;
            if (dt >= 0) feed_imu(s, front_t, k1[0], k0 + 1, k0 + 4, k1 + 1, k1 + 4);
            front++;                       /* erase(begin()) */
            front_t = stream[front * 7];
        }
        double dt_f = updatetime - front_t;
        if (dt_f > 0) {
            const double *k0 = stream + front * 7;
            feed_imu(s, front_t, updatetime, k0 + 1, k0 + 4, k0 + 1, k0 + 4);
            front_t = updatetime;          /* imu_times.at(0) = updatetime */
        }
        state_export(s, out + u);
    }
    free(s);
}

typedef struct {
    const cpi_oracle_params *prm; long w0, w1; int n;
    const double *knots, *lin, *q; cpi_oracle_out *out;
} mt_job;

static void *mt_worker(void *p) {
    mt_job *j = (mt_job *)p;
    for (long w = j->w0; w < j->w1; w++)
        run_window(j->prm, j->n, j->knots + (size_t)w * (j->n + 1) * 7, j->lin + (size_t)w * 6,
                   j->q ? j->q + (size_t)w * 4 : NULL, j->out + w, NULL);
    return NULL;
}

void cpi_oracle_batch_mt(const cpi_oracle_params *prm, long W, int n, const double *knots, const double *lin,
                         const double *q_k_lin, cpi_oracle_out *out, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    mt_job *jobs = (mt_job *)malloc(sizeof(mt_job) * nthreads);
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (mt_job){ prm, W * t / nthreads, W * (t + 1) / nthreads, n, knots, lin, q_k_lin, out };
        pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
}

/* ---------------------------------------------------------------- factors */
static void from_colmajor3(const double *A, double *out) { mt3(A, out); }

/* (q_w I - skew(q_v)) */
static void qL(const double *q, double sign, double *M) {
    double S[9];
    skew_x(q, S);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[i * 3 + j] = q[3] * (i == j ? 1.0 : 0.0) + sign * S[i * 3 + j];
}

static void factor_eval(int v2, const cpi_oracle_factor *f, const double *xi, const double *xj,
                        double *err, double *H1, double *H2) {
    const double *q_GtoK = xi, *bg_K = xi + 4, *v_K = xi + 7, *ba_K = xi + 10, *p_K = xi + 13;
    const double *q_GtoK1 = xj, *bg_K1 = xj + 4, *v_K1 = xj + 7, *ba_K1 = xj + 10, *p_K1 = xj + 13;
    double J_q[9], J_beta[9], J_alpha[9], H_beta[9], H_alpha[9], O_beta[9], O_alpha[9];
    from_colmajor3(f->J_q, J_q); from_colmajor3(f->J_beta, J_beta); from_colmajor3(f->J_alpha, J_alpha);
    from_colmajor3(f->H_beta, H_beta); from_colmajor3(f->H_alpha, H_alpha);
    from_colmajor3(f->O_beta, O_beta); from_colmajor3(f->O_alpha, O_alpha);
    double dt = f->deltatime;

    double dbg[3], dba[3];
    for (int i = 0; i < 3; i++) { dbg[i] = bg_K[i] - f->bg_lin[i]; dba[i] = ba_K[i] - f->ba_lin[i]; }

    /* ImuFactorCPIv1.cpp:57-64 */
    double arg[3], ExpB[9], q_b[4];
    mv3(J_q, dbg, arg);
    for (int i = 0; i < 3; i++) arg[i] = -arg[i];
    Exp_so3(arg, ExpB);
    rot_2_quat(ExpB, q_b);

    double qi[4], q_n[4], q_rminus[4], q_r[4], q_m[4], q_kR[4] = {0, 0, 0, 1}, dthk[3] = {0, 0, 0};
    quat_inv(q_GtoK, qi);          quat_multiply(q_GtoK1, qi, q_n);
    quat_inv(f->q_KtoK1, qi);      quat_multiply(q_n, qi, q_rminus);
    quat_multiply(q_rminus, q_b, q_r);
    quat_inv(q_b, qi);             quat_multiply(qi, f->q_KtoK1, q_m);
    if (v2) {
        quat_inv(f->q_K_lin, qi);  quat_multiply(q_GtoK, qi, q_kR);
        for (int i = 0; i < 3; i++) dthk[i] = 2 * q_kR[i];
    }

    double Rk[9], pa[3], pb[3], Ra[3], Rb[3];
    quat_2_Rot(q_GtoK, Rk);
    if (!v2) {
        for (int i = 0; i < 3; i++) {
            pa[i] = p_K1[i] - p_K[i] - v_K[i] * dt + 0.5 * f->grav[i] * pow(dt, 2);
            pb[i] = v_K1[i] - v_K[i] + f->grav[i] * dt;
        }
    } else {
        for (int i = 0; i < 3; i++) {
            pa[i] = p_K1[i] - p_K[i] - v_K[i] * dt;
            pb[i] = v_K1[i] - v_K[i];
        }
    }
    mv3(Rk, pa, Ra);
    mv3(Rk, pb, Rb);
    double alphahat[3], betahat[3], t1[3], t2[3], t3[3];
    mv3(J_alpha, dbg, t1); mv3(H_alpha, dba, t2); mv3(O_alpha, dthk, t3);
    for (int i = 0; i < 3; i++) alphahat[i] = Ra[i] - t1[i] - t2[i] - (v2 ? t3[i] : 0.0);
    mv3(J_beta, dbg, t1); mv3(H_beta, dba, t2); mv3(O_beta, dthk, t3);
    for (int i = 0; i < 3; i++) betahat[i] = Rb[i] - t1[i] - t2[i] - (v2 ? t3[i] : 0.0);

    for (int i = 0; i < 3; i++) {
        err[0 + i] = 2 * q_r[i];
        err[3 + i] = bg_K1[i] - bg_K[i];
        err[6 + i] = betahat[i] - f->beta[i];
        err[9 + i] = ba_K1[i] - ba_K[i];
        err[12 + i] = alphahat[i] - f->alpha[i];
    }

    if (H1) {
        double Hi[225];
        memset(Hi, 0, sizeof Hi);
        double A[9], B[9], AB[9], T[9], S[9], I[9];
        eye3(I);
        /* H_theta(0,0) */
        qL(q_n, -1.0, A); qL(q_m, -1.0, B); mm3(A, B, AB);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T[i * 3 + j] = -(AB[i * 3 + j] + q_n[i] * q_m[j]);
        set_block(15, Hi, 0, 0, T, 1.0);
        /* H_theta(6,0), (12,0) */
        skew_x(Rb, S);
        if (v2) { qL(q_kR, +1.0, A); mm3(O_beta, A, T); for (int i = 0; i < 9; i++) S[i] -= T[i]; }
        set_block(15, Hi, 6, 0, S, 1.0);
        skew_x(Ra, S);
        if (v2) { qL(q_kR, +1.0, A); mm3(O_alpha, A, T); for (int i = 0; i < 9; i++) S[i] -= T[i]; }
        set_block(15, Hi, 12, 0, S, 1.0);
        /* H_biasg */
        qL(q_rminus, -1.0, A); mm3(A, J_q, T);
        set_block(15, Hi, 0, 3, T, 1.0);
        set_block(15, Hi, 3, 3, I, -1.0);
        set_block(15, Hi, 6, 3, J_beta, -1.0);
        set_block(15, Hi, 12, 3, J_alpha, -1.0);
        /* H_velocity */
        set_block(15, Hi, 6, 6, Rk, -1.0);
        set_block(15, Hi, 12, 6, Rk, -dt);
        /* H_biasa */
        set_block(15, Hi, 6, 9, H_beta, -1.0);
        set_block(15, Hi, 9, 9, I, -1.0);
        set_block(15, Hi, 12, 9, H_alpha, -1.0);
        /* H_position */
        set_block(15, Hi, 12, 12, Rk, -1.0);
        for (int i = 0; i < 15; i++) for (int j = 0; j < 15; j++) H1[j * 15 + i] = Hi[i * 15 + j];
    }
    if (H2) {
        double Hj[225], A[9], I[9];
        memset(Hj, 0, sizeof Hj);
        eye3(I);
        qL(q_r, +1.0, A);
        set_block(15, Hj, 0, 0, A, 1.0);
        set_block(15, Hj, 3, 3, I, 1.0);
        set_block(15, Hj, 6, 6, Rk, 1.0);
        set_block(15, Hj, 9, 9, I, 1.0);
        set_block(15, Hj, 12, 12, Rk, 1.0);
        for (int i = 0; i < 15; i++) for (int j = 0; j < 15; j++) H2[j * 15 + i] = Hj[i * 15 + j];
    }
}

void cpi_oracle_factor_v1(const cpi_oracle_factor *f, const double *xi, const double *xj, double *err, double *H1, double *H2) {
    factor_eval(0, f, xi, xj, err, H1, H2);
}
void cpi_oracle_factor_v2(const cpi_oracle_factor *f, const double *xi, const double *xj, double *err, double *H1, double *H2) {
    factor_eval(1, f, xi, xj, err, H1, H2);
}

/* F factors (records of 87 doubles, the field order of cpi_oracle_factor), spread over nthreads pthreads: the CPU leg
 * bench.py times beside the re-linearisation sweep ("port": the reference's factor TUs need GTSAM and cannot be built). */
typedef struct { int model; long f0, f1; const double *rec, *xi, *xj; double *err, *H1, *H2; } fb_job;
static void *fb_worker(void *p) {
    fb_job *j = (fb_job *)p;
    for (long f = j->f0; f < j->f1; f++)
        factor_eval(j->model == 2, (const cpi_oracle_factor *)(j->rec + f * 87), j->xi + f * 16, j->xj + f * 16,
                    j->err + f * 15, j->H1 ? j->H1 + f * 225 : NULL, j->H2 ? j->H2 + f * 225 : NULL);
    return NULL;
}
void cpi_oracle_factor_batch_mt(int model, long F, const double *rec, const double *xi, const double *xj, double *err,
                                double *H1, double *H2, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    fb_job *jobs = (fb_job *)malloc(sizeof(fb_job) * nthreads);
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (fb_job){ model, F * t / nthreads, F * (t + 1) / nthreads, rec, xi, xj, err, H1, H2 };
        pthread_create(&th[t], NULL, fb_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
}

/* GraphSolver_IMU.cpp:263-281 / 289-307 */
void cpi_oracle_predict(int model, const cpi_oracle_factor *f, const double *xi, double *xj) {
    const double *q_GtoK = xi, *bg_K = xi + 4, *v_K = xi + 7, *ba_K = xi + 10, *p_K = xi + 13;
    double qi[4], Rinv[9], rb[3], ra[3];
    quat_multiply(f->q_KtoK1, q_GtoK, xj);
    quat_inv(q_GtoK, qi);
    quat_2_Rot(qi, Rinv);
    mv3(Rinv, f->beta, rb);
    mv3(Rinv, f->alpha, ra);
    double dt = f->deltatime;
    for (int i = 0; i < 3; i++) {
        xj[4 + i] = bg_K[i];
        xj[10 + i] = ba_K[i];
        if (model == 1) {
            xj[7 + i] = v_K[i] - f->grav[i] * dt + rb[i];
            xj[13 + i] = p_K[i] + v_K[i] * dt - 0.5 * f->grav[i] * pow(dt, 2) + ra[i];
        } else {
            xj[7 + i] = v_K[i] + rb[i];
            xj[13 + i] = p_K[i] + v_K[i] * dt + ra[i];
        }
    }
}

/* JPLNavState.cpp:37-71 */
void cpi_oracle_retract(const double *x, const double *xi, double *xout) {
    double n = norm3(xi);
    double dq[4];
    for (int i = 0; i < 3; i++) dq[i] = ((sin(n / 2) / n)) * xi[i];
    dq[3] = cos(n / 2);
    double nn = sqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3]);
    for (int i = 0; i < 4; i++) dq[i] /= nn;
    if (dq[3] < 0) for (int i = 0; i < 4; i++) dq[i] = -dq[i];
    nn = sqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3]);
    if (isnan(nn)) { dq[0] = dq[1] = dq[2] = 0; dq[3] = 1.0; }
    quat_multiply(dq, x, xout);
    for (int i = 0; i < 12; i++) xout[4 + i] = x[4 + i] + xi[3 + i];
}

/* JPLNavState.cpp:80-88 */
void cpi_oracle_local(const double *x, const double *other, double *xi) {
    double qi[4], qd[4];
    quat_inv(x, qi);
    quat_multiply(other, qi, qd);
    for (int i = 0; i < 3; i++) xi[i] = 2 * qd[i];
    for (int i = 0; i < 12; i++) xi[3 + i] = other[4 + i] - x[4 + i];
}
