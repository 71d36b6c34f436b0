/*
 * ref_shim.cpp -- thin extern "C" driver around the REFERENCE's own preintegrator headers.
 *
 * TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it #includes
 * cpi/CpiV1.h and cpi/CpiV2.h (and through them utils/quat_ops.h and the vendored Eigen 3.2.10)
 * from where they lie under /root/reference at build time (see Makefile) and exposes them with
 * the same signature as cpi_oracle_window(), so the C restatement can be validated against the
 * real thing and the real thing can be timed as the CPU baseline ("kind": "reference").
 * The output lives in oracle/_ref/ (git-ignored, never committed).
 *
 * The factors (ImuFactorCPIv1/v2.cpp) need GTSAM + Boost, absent from this image, and are NOT
 * built here: no stand-in headers are written for them.  What CAN be pinned of evaluateError / predict / retract
 * is: every non-trivial primitive those bodies call lives in utils/quat_ops.h (rot_2_quat :45, skew_x :92,
 * quat_2_Rot :104, quat_multiply :115, Exp :145, Inv :190), header-only on the vendored Eigen -- cpi_ref_quat_ops()
 * below exposes the reference's OWN functions so that the restatement's helpers and the device helpers are checked
 * against them (tests/golden/quat_ops.npz, tests/test_quat_ops.py, tests/test_gpu_quat_ops.py).
 */
#include "cpi/CpiV1.h"
#include "cpi/CpiV2.h"
#include "cpi_oracle.h"
#include <thread>
#include <vector>

typedef Eigen::Matrix<double, 3, 1> V3;
typedef Eigen::Matrix<double, 4, 1> V4;

template <class CPI>
static void run(CPI &cpi, int n, const double *knots, const double *lin, const double *q_k_lin,
                const double *grav) {
    V3 bw(lin[0], lin[1], lin[2]), ba(lin[3], lin[4], lin[5]);
    V4 qk = V4::Zero();
    if (q_k_lin) qk << q_k_lin[0], q_k_lin[1], q_k_lin[2], q_k_lin[3];
    cpi.setLinearizationPoints(bw, ba, qk, V3(grav[0], grav[1], grav[2]));
    cpi.q_k2tau << 0, 0, 0, 1; /* uninitialised in the reference until the first step */
    for (int i = 0; i < n; i++) {
        const double *k0 = knots + 7 * i, *k1 = knots + 7 * (i + 1);
        double dt = k1[0] - k0[0];
        if (dt >= 0) /* GraphSolver_IMU.cpp:52 */
            cpi.feed_IMU(k0[0], k1[0], V3(k0[1], k0[2], k0[3]), V3(k0[4], k0[5], k0[6]),
                         V3(k1[1], k1[2], k1[3]), V3(k1[4], k1[5], k1[6]));
    }
}

template <class CPI>
static void export_base(const CPI &c, cpi_oracle_out *o) {
    o->DT = c.DT;
    for (int i = 0; i < 3; i++) { o->alpha[i] = c.alpha_tau(i); o->beta[i] = c.beta_tau(i); }
    for (int i = 0; i < 4; i++) o->q[i] = c.q_k2tau(i);
    /* Eigen fixed-size matrices are column-major: copy raw storage */
    std::copy(c.R_k2tau.data(), c.R_k2tau.data() + 9, o->R);
    std::copy(c.J_q.data(), c.J_q.data() + 9, o->J_q);
    std::copy(c.J_a.data(), c.J_a.data() + 9, o->J_a);
    std::copy(c.J_b.data(), c.J_b.data() + 9, o->J_b);
    std::copy(c.H_a.data(), c.H_a.data() + 9, o->H_a);
    std::copy(c.H_b.data(), c.H_b.data() + 9, o->H_b);
    std::copy(c.P_meas.data(), c.P_meas.data() + 225, o->P);
    for (int i = 0; i < 9; i++) { o->O_a[i] = 0; o->O_b[i] = 0; }
}

extern "C" void cpi_ref_window(const cpi_oracle_params *prm, int n, const double *knots,
                               const double *lin, const double *q_k_lin, cpi_oracle_out *out) {
    if (prm->model == 1) {
        CpiV1 cpi(prm->sigma_w, prm->sigma_wb, prm->sigma_a, prm->sigma_ab, prm->imu_avg != 0);
        run(cpi, n, knots, lin, q_k_lin, prm->grav);
        export_base(cpi, out);
    } else {
        CpiV2 cpi(prm->sigma_w, prm->sigma_wb, prm->sigma_a, prm->sigma_ab, prm->imu_avg != 0);
        cpi.state_transition_jacobians = prm->state_transition_jacobians != 0;
        run(cpi, n, knots, lin, q_k_lin, prm->grav);
        export_base(cpi, out);
        std::copy(cpi.O_a.data(), cpi.O_a.data() + 9, out->O_a);
        std::copy(cpi.O_b.data(), cpi.O_b.data() + 9, out->O_b);
    }
}

extern "C" void cpi_ref_batch(const cpi_oracle_params *prm, long W, int n, const double *knots,
                              const double *lin, const double *q_k_lin, cpi_oracle_out *out) {
    for (long w = 0; w < W; w++)
        cpi_ref_window(prm, n, knots + (size_t)w * (n + 1) * 7, lin + (size_t)w * 6,
                       q_k_lin ? q_k_lin + (size_t)w * 4 : nullptr, out + w);
}

extern "C" void cpi_ref_batch_mt(const cpi_oracle_params *prm, long W, int n, const double *knots,
                                 const double *lin, const double *q_k_lin, cpi_oracle_out *out,
                                 int nthreads) {
    if (nthreads < 1) nthreads = 1;
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) {
        long w0 = W * t / nthreads, w1 = W * (t + 1) / nthreads;
        th.emplace_back([=]() {
            for (long w = w0; w < w1; w++)
                cpi_ref_window(prm, n, knots + (size_t)w * (n + 1) * 7, lin + (size_t)w * 6,
                               q_k_lin ? q_k_lin + (size_t)w * 4 : nullptr, out + w);
        });
    }
    for (auto &t : th) t.join();
}

/* The reference's own quat_ops.h helpers, one call per item.  Matrices cross this interface ROW-major.
 *   op 0 rot_2_quat    in 9  -> out 4      op 1 skew_x         in 3 -> out 9      op 2 quat_2_Rot  in 4 -> out 9
 *   op 3 quat_multiply in 4+4 -> out 4     op 4 Exp            in 3 -> out 9      op 5 Inv         in 4 -> out 4 */
extern "C" int cpi_ref_quat_ops(int op, long n, const double *in, double *out) {
    typedef Eigen::Matrix<double, 3, 3, Eigen::RowMajor> M3r;
    for (long k = 0; k < n; k++) {
        switch (op) {
            case 0: {
                Eigen::MatrixXd R = Eigen::Map<const M3r>(in + 9 * k);
                V4 q = rot_2_quat(R);
                for (int i = 0; i < 4; i++) out[4 * k + i] = q(i);
            } break;
            case 1: {
                Eigen::MatrixXd S = skew_x(V3(in[3 * k], in[3 * k + 1], in[3 * k + 2]));
                Eigen::Map<M3r>(out + 9 * k) = S;
            } break;
            case 2: {
                V4 q; q << in[4 * k], in[4 * k + 1], in[4 * k + 2], in[4 * k + 3];
                Eigen::MatrixXd R = quat_2_Rot(q);
                Eigen::Map<M3r>(out + 9 * k) = R;
            } break;
            case 3: {
                V4 q, p;
                q << in[8 * k], in[8 * k + 1], in[8 * k + 2], in[8 * k + 3];
                p << in[8 * k + 4], in[8 * k + 5], in[8 * k + 6], in[8 * k + 7];
                V4 r = quat_multiply(q, p);
                for (int i = 0; i < 4; i++) out[4 * k + i] = r(i);
            } break;
            case 4: {
                Eigen::Matrix<double, 3, 3> R = Exp(V3(in[3 * k], in[3 * k + 1], in[3 * k + 2]));
                Eigen::Map<M3r>(out + 9 * k) = R;
            } break;
            case 5: {
                V4 q; q << in[4 * k], in[4 * k + 1], in[4 * k + 2], in[4 * k + 3];
                V4 r = Inv(q);
                for (int i = 0; i < 4; i++) out[4 * k + i] = r(i);
            } break;
            default: return 1;
        }
    }
    return 0;
}
