/*
 * cpi_oracle.h -- CPU restatement of the rpng/cpi continuous-preintegration hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load or call it, and only
 * as the checker / the CPU baseline.  The shipped path is the HIP library (libcpi_amd.so).
 *
 * Parity status
 *   - cpi_oracle_window (CpiV1 / CpiV2 feed_IMU): PINNED against the reference's own headers,
 *     compiled unchanged from /root/reference into oracle/_ref/libcpi_ref.so (see Makefile,
 *     ref_shim.cpp) and against the committed golden vectors in tests/golden/.
 *   - cpi_oracle_quat_ops (the quat_ops.h helpers every function below is built from: rot_2_quat,
 *     skew_x, quat_2_Rot, quat_multiply, Exp, Inv): PINNED to the reference's own functions, compiled
 *     into oracle/_ref (ref_shim.cpp: cpi_ref_quat_ops) and to tests/golden/quat_ops.npz (<= 1e-15).
 *   - cpi_oracle_factor_v1/v2 (ImuFactorCPIv1/v2::evaluateError), cpi_oracle_predict,
 *     cpi_oracle_retract: the BLOCK ASSEMBLY is PARITY UNPINNED.  The reference translation units
 *     need GTSAM + Boost, which are not in this image, and stand-in headers are not allowed; the
 *     reference has no tests or golden vectors for them.  They are a line-by-line restatement on top
 *     of the pinned helpers, cross-checked by finite differences against JPLNavState::retract
 *     semantics (tests/test_factor_oracle.py; on the device: tests/test_gpu_quat_ops.py).
 *   - cpi_oracle_forster_window (the GTSAM "Forster discrete" comparator): PARITY UNPINNED, GTSAM is absent;
 *     see the header of forster_oracle.c for what it restates and what the tests pin instead.
 *
 * Storage convention at this API: every 3x3 / 15x15 matrix is COLUMN-MAJOR (Eigen default),
 * quaternions are JPL [x y z w], state/tangent order is [theta b_g v b_a p].
 */
#ifndef CPI_ORACLE_H
#define CPI_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    double sigma_w, sigma_wb, sigma_a, sigma_ab; /* CpiBase.h:52 */
    double grav[3];                              /* CpiBase.h:118 */
    int model;                                   /* 1 = CpiV1, 2 = CpiV2, 3 = Forster comparator (forster_oracle.c) */
    int imu_avg;                                 /* CpiBase.h:95 */
    int state_transition_jacobians;              /* CpiV2.h:58 (V2 only) */
} cpi_oracle_params;

typedef struct {
    double DT, alpha[3], beta[3], q[4];          /* CpiBase.h:99-102 */
    double R[9];                                 /* CpiBase.h:103 */
    double J_q[9], J_a[9], J_b[9], H_a[9], H_b[9]; /* CpiBase.h:106-110 */
    double O_a[9], O_b[9];                       /* CpiV2.h:62-63 */
    double P[225];                               /* CpiBase.h:124 */
} cpi_oracle_out;

/* One window.  knots = (n+1) records {t, w[3], a[3]}; interval i = feed_IMU(t_i, t_{i+1},
 * w_i, a_i, w_{i+1}, a_{i+1}) and is skipped when t_{i+1}-t_i < 0 (GraphSolver_IMU.cpp:50-53).
 * lin = {b_w_lin[3], b_a_lin[3]}; q_k_lin may be NULL for model 1. */
void cpi_oracle_window(const cpi_oracle_params *prm, int n, const double *knots,
                       const double *lin, const double *q_k_lin, cpi_oracle_out *out);

/* Per-sample trace of one window (test localisation): after every interval i writes
 * trace[i] = full cpi_oracle_out snapshot. */
void cpi_oracle_window_trace(const cpi_oracle_params *prm, int n, const double *knots,
                             const double *lin, const double *q_k_lin, cpi_oracle_out *trace);

/* W windows, fixed n, dense layout knots[W][n+1][7], lin[W][6], q_k_lin[W][4] (or NULL). */
void cpi_oracle_batch(const cpi_oracle_params *prm, long W, int n, const double *knots,
                      const double *lin, const double *q_k_lin, cpi_oracle_out *out);

/* Window assembly exactly as GraphSolver::createimufactor_cpi_v1/v2 does it (GraphSolver_IMU.cpp:50-69): ONE
 * IMU stream (K knots {t,w,a}) is consumed front to back by U successive update times; each update
 * integrates whole intervals while imu_times[1] <= updatetime, then the partial tail interval
 * [imu_times[0], updatetime] with the front reading repeated, and overwrites imu_times[0] = updatetime.
 * lin [U][6], q_k_lin [U][4] (may be NULL for model 1); out [U]. */
void cpi_oracle_stream(const cpi_oracle_params *prm, long K, const double *stream, long U,
                       const double *update_times, const double *lin, const double *q_k_lin,
                       cpi_oracle_out *out);

/* Same, spread over nthreads pthreads (CPU baseline on all host cores). */
void cpi_oracle_batch_mt(const cpi_oracle_params *prm, long W, int n, const double *knots,
                         const double *lin, const double *q_k_lin, cpi_oracle_out *out,
                         int nthreads);

/* Factor measurement record (what the ImuFactorCPIv1/v2 constructors copy,
 * ImuFactorCPIv1.h:78-100, ImuFactorCPIv2.h:82-102). */
typedef struct {
    double alpha[3], beta[3], q_KtoK1[4];
    double ba_lin[3], bg_lin[3];
    double J_q[9], J_beta[9], J_alpha[9], H_beta[9], H_alpha[9];
    double deltatime, grav[3];
    double q_K_lin[4], O_beta[9], O_alpha[9]; /* v2 only */
} cpi_oracle_factor;

/* state = JPLNavState 16 doubles [q(4) bg(3) v(3) ba(3) p(3)] (JPLNavState.h:62-66).
 * err[15]; H1,H2 15x15 column-major, may be NULL. */
void cpi_oracle_factor_v1(const cpi_oracle_factor *f, const double *xi, const double *xj,
                          double *err, double *H1, double *H2);
void cpi_oracle_factor_v2(const cpi_oracle_factor *f, const double *xi, const double *xj,
                          double *err, double *H1, double *H2);

/* The same for F factors: rec [F][87] (this struct's fields as consecutive doubles), xi / xj [F][16]; nthreads pthreads. */
void cpi_oracle_factor_batch_mt(int model, long F, const double *rec, const double *xi, const double *xj, double *err,
                                double *H1, double *H2, int nthreads);

/* GraphSolver_IMU.cpp:263-281 (model 1) / 289-307 (model 2). */
void cpi_oracle_predict(int model, const cpi_oracle_factor *f, const double *xi, double *xj);

/* Forster discrete comparator (GraphSolver::createimufactor_discrete, GraphSolver_IMU.cpp:141-232): GTSAM's
 * PreintegratedCombinedMeasurements restated (PARITY UNPINNED, see forster_oracle.c), outputs already converted
 * the way the call site does it (q = rot_2_quat(deltaRij^T), J_q = -delRdelBiasOmega, covariance block-swapped
 * into [theta b_g v b_a p]).  imu_avg / q_k_lin / grav are not used.  Also reached through cpi_oracle_window /
 * cpi_oracle_batch with prm->model == 3. */
void cpi_oracle_forster_window(const cpi_oracle_params *prm, int n, const double *knots, const double *lin,
                               cpi_oracle_out *out);
/* gtsam::NavState::retract / ::update restated (test hooks); state15 = {nRb row-major 9, n_t 3, n_v 3};
 * A 9x9, B 9x3, C 9x3 row-major (any may be NULL). */
void cpi_oracle_navstate_retract(const double *state15, const double *xi9, double *out15);
void cpi_oracle_navstate_update(const double *state15, const double *acc, const double *om, double dt,
                                double *out15, double *A81, double *B27, double *C27);

/* The restated quat_ops.h helpers (quat_ops.h:45-197), one call per item; matrices ROW-major at this interface.
 *   op 0 rot_2_quat in 9 -> out 4 | 1 skew_x in 3 -> out 9 | 2 quat_2_Rot in 4 -> out 9 | 3 quat_multiply in 4+4 -> out 4
 *   op 4 Exp in 3 -> out 9 | 5 Inv in 4 -> out 4.      PINNED: equal to the reference's own functions (compiled into
 * oracle/_ref by ref_shim.cpp: cpi_ref_quat_ops) on tests/golden/quat_ops.npz.  Returns 0, or 1 for an unknown op. */
int cpi_oracle_quat_ops(int op, long n, const double *in, double *out);

/* JPLNavState::retract (JPLNavState.cpp:37-71) and localCoordinates (:80-88). */
void cpi_oracle_retract(const double *x, const double *xi15, double *xout);
void cpi_oracle_local(const double *x, const double *other, double *xi15);

#ifdef __cplusplus
}
#endif
#endif
