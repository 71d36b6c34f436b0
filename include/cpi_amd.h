/*
 * cpi_amd.h -- C-ABI of the MI355X-native batched continuous-preintegration engine.
 *
 * This is the drop-in boundary for the rpng/cpi hot path.  The reference has no FFI layer: the
 * boundary there is two C++ class interfaces compiled into the caller,
 *   - the preintegrator  CpiBase / CpiV1 / CpiV2   (cpi_compare/src/cpi/CpiBase.h:40-145,
 *     CpiV1.h:62, CpiV2.h:84): ctor(sigmas, imu_avg), setLinearizationPoints(), feed_IMU() called
 *     once per IMU interval (GraphSolver_IMU.cpp:50-69), results read from public members
 *     (GraphSolver_IMU.cpp:74-75,129-130);
 *   - the factor  ImuFactorCPIv1/v2::evaluateError(state_i, state_j, H1, H2)
 *     (cpi_compare/src/gtsam/ImuFactorCPIv1.h:139 / .cpp:37, ImuFactorCPIv2.h:151 / .cpp:38),
 *     called by GTSAM once per factor per re-linearisation.
 * The entry points below are the batched equivalents a binding of those two interfaces would call
 * (see INTEGRATION.md for the reference-side stub).  Plain pointers and sizes only.
 *
 * Conventions
 *   - all arithmetic is IEEE double; all pointers are DEVICE pointers unless the name says _host;
 *   - every 3x3 / 15x15 matrix is COLUMN-MAJOR (Eigen's default, so Eigen::Map works unchanged);
 *   - quaternions are JPL [x y z w] with w >= 0 (quat_ops.h:80-82);
 *   - error-state / tangent order is [theta b_g v b_a p] (ImuFactorCPIv1.cpp:80);
 *   - JPLNavState is 16 doubles [q(4) b_g(3) v(3) b_a(3) p(3)] (JPLNavState.h:62-66);
 *   - calls on one cpi_ctx are ordered on its HIP stream and return without synchronising;
 *     distinct contexts (one per GPU / per stream) are independent.  No global state.
 */
#ifndef CPI_AMD_H
#define CPI_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPI_ABI_VERSION 3   /* 3 (round 6): cpi_outputs gained a 13th field, P_sym (the covariance as its packed upper triangle) -- a binding
                               built against version 2 passes a 12-field struct and must be rebuilt; new symbols
                               cpi_sqrt_information_packed_batch, cpi_factor_eval_whitened_tri_batch, cpi_factor_hessian_tri_batch,
                               cpi_group_gather_chunk, cpi_shard_chunk_bounds; this number also records round 4's workspace-contract
                               change of cpi_preintegrate_stream (below), which version 2 carried only in prose.
                               2: state count S in the factor / predict entries; device-set entries (cpi_group_*);
                               additions within 2 (new symbols only): tiled layout entries, cpi_host_alloc / _free;
                               round 3: cpi_preintegrate_stream (+ _workspace_bytes, _counts), cpi_tile_windows,
                               cpi_assemble_tiles, cpi_preintegrate_tiled_batch_host,
                               cpi_outputs_slab_doubles / _bind_slab, cpi_group_last_gather_messages,
                               cpi_preintegrate_stream_host;
                               round 4, a CONTRACT change without a new symbol: after cpi_preintegrate_stream the workspace
                               holds the true interval counts (cpi_stream_counts) and NOTHING ELSE a caller may read -- a
                               mean-only request runs no cut kernel, so the first / tstart / tend records round 3 left there
                               are no longer written (they were never declared) */

enum { CPI_OK = 0, CPI_ERR_INVALID = 1, CPI_ERR_HIP = 2, CPI_ERR_NO_DEVICE = 3, CPI_ERR_RCCL = 4 };
enum {
    CPI_MODEL_V1 = 1, CPI_MODEL_V2 = 2,
    /* cpi_preintegrate_batch only: the "Forster discrete" comparator, i.e. what GraphSolver::createimufactor_discrete
     * (GraphSolver_IMU.cpp:141-232) gets from GTSAM's PreintegratedCombinedMeasurements (manifold preintegration),
     * already converted the way that call site does it: alpha = deltaPij, beta = deltaVij,
     * q = rot_2_quat(deltaRij^T), J_q = -delRdelBiasOmega, J_a / J_b = delP / delV delBiasOmega,
     * H_a / H_b = delP / delV delBiasAcc, P = preintMeasCov with blocks 1 and 4 swapped (swapcovariance :240-254)
     * = order [theta b_g v b_a p].  Reading i is held over [t_i, t_i+1] (no averaging: imu_avg, q_k_lin, grav,
     * lanes_per_window are ignored).  The result is the measurement of an ImuFactorCPIv1 (:227-231): evaluate it
     * with model CPI_MODEL_V1.  GTSAM is not part of the reference tree: parity of this model is UNPINNED
     * (oracle/forster_oracle.c). */
    CPI_MODEL_FORSTER = 3
};

typedef struct cpi_ctx cpi_ctx;

/* Replaces: CpiBase ctor arguments (CpiBase.h:52), the imu_avg flag (CpiBase.h:95), the
 * state_transition_jacobians flag (CpiV2.h:58) and the gravity passed to setLinearizationPoints
 * (CpiBase.h:73-80; one global gravity per batch, as in Config.h / GraphSolver_IMU.cpp:44). */
typedef struct {
    double sigma_w, sigma_wb, sigma_a, sigma_ab;
    double grav[3];
    int32_t model;                       /* CPI_MODEL_V1 | CPI_MODEL_V2 | CPI_MODEL_FORSTER (preintegration only) */
    int32_t imu_avg;                     /* 0 / 1 */
    int32_t state_transition_jacobians;  /* model 2 only; reference default 1 */
    int32_t lanes_per_window;            /* mean kernel: 0 = auto, else 1,2,3,4,5,6,8,12,16,32,64 (tuning knob;
                                            ignored by the covariance kernel and by model 2 with analytic Jacobians).
                                            cpi_preintegrate_tiled_batch reads it as WAVEFRONTS PER TILE: 0 = auto, else 1..8 */
} cpi_params;

/* Replaces: the public result members of CpiBase / CpiV2 (CpiBase.h:99-124, CpiV2.h:62-63).
 * Structure-of-arrays over the W windows of a batch; any pointer may be NULL = "not wanted":
 *   DT, alpha, beta, q all NULL      -> means are not written
 *   J_q ... O_b all NULL             -> bias / orientation Jacobians are not computed
 *   P and P_sym both NULL            -> the covariance recursion is skipped (mean-only kernel) */
typedef struct {
    double *DT;     /* [W]       CpiBase::DT        */
    double *alpha;  /* [W][3]    alpha_tau          */
    double *beta;   /* [W][3]    beta_tau           */
    double *q;      /* [W][4]    q_k2tau            */
    double *J_q;    /* [W][9]    orientation wrt b_w */
    double *J_a;    /* [W][9]    alpha wrt b_w      */
    double *J_b;    /* [W][9]    beta wrt b_w       */
    double *H_a;    /* [W][9]    alpha wrt b_a      */
    double *H_b;    /* [W][9]    beta wrt b_a       */
    double *O_a;    /* [W][9]    alpha wrt q_k_lin (model 2) */
    double *O_b;    /* [W][9]    beta wrt q_k_lin  (model 2) */
    double *P;      /* [W][225]  P_meas, dense column-major (the drop-in form: Eigen::Map<Matrix<double,15,15>>) */
    double *P_sym;  /* [W][120]  P_meas as its packed upper triangle (CPI_TRI_INDEX below; P_meas is symmetric -- the
                                 reference itself asserts it, CpiV1.h:352-353 -- so the dense form carries 105 redundant doubles
                                 = 840 of the 2 248 bytes a model-1 window writes with everything out).  P and P_sym are independent: either, both or
                                 neither; the entries of P_sym are bit for bit the entries (i, j), i <= j, of P */
} cpi_outputs;

/* Packed triangles (ABI 3).  A symmetric 15 x 15 matrix (P_meas) is stored as its upper triangle, an upper-triangular one (the
 * square-root information R) as its non-zero part, both COLUMN by column -- LAPACK's packed 'U' order, the order
 * cpi_factor_hessian_batch already writes its 31 x 31 triangle in:
 *     entry (i, j), i <= j, at  CPI_TRI_INDEX(i, j) = i + j (j + 1) / 2,      120 doubles = 960 bytes instead of 1 800.
 * Column j is the run [j (j + 1) / 2, j (j + 1) / 2 + j].  INTEGRATION.md section 4a shows the Eigen one-liners
 * (selfadjointView<Upper> / triangularView<Upper>) a GTSAM-side binding unpacks them with; cpi_amd.unpack_sym / unpack_tri /
 * pack_sym are the Python mirrors. */
#define CPI_TRI_DOUBLES 120
#define CPI_TRI_INDEX(i, j) ((i) + (j) * ((j) + 1) / 2)

/* device < 0: use the current HIP device.  stream: a hipStream_t (NULL = the default stream). */
int cpi_ctx_create(int device, void *stream, cpi_ctx **out);
void cpi_ctx_destroy(cpi_ctx *ctx);
/* Later calls on ctx are issued on `stream` (a hipStream_t of the context's device).  The caller orders the two streams. */
int cpi_ctx_set_stream(cpi_ctx *ctx, void *stream);
const char *cpi_last_error(const cpi_ctx *ctx); /* ctx may be NULL: last error of a failed create */
int cpi_abi_version(void);
const char *cpi_build_id(void);   /* sha256[:16] of the sources this library was built from (cpi_amd/build.py: source_id) */
int cpi_ctx_synchronize(cpi_ctx *ctx);

/* Replaces: the per-window loop  CpiV{1,2} cpi(...); cpi.setLinearizationPoints(...);
 *           while (...) cpi.feed_IMU(t0,t1,w0,a0,w1,a1);   (GraphSolver_IMU.cpp:43-69, 97-124).
 *
 * knots   IMU knot records {t, w[3], a[3]} (7 doubles).  Interval i of a window is
 *         feed_IMU(t_i, t_{i+1}, w_i, a_i, w_{i+1}, a_{i+1}); intervals with t_{i+1}-t_i <= 0 are
 *         skipped exactly like the reference (dt==0: CpiV1.h:72; dt<0: GraphSolver_IMU.cpp:52).
 *         A tail interval [t_last, updatetime] is expressed by a final knot
 *         {updatetime, w_last, a_last} (GraphSolver_IMU.cpp:64-69).  A knot whose t is NaN is a
 *         separator: both intervals touching it are skipped, so non-chained feed_IMU calls
 *         (the reference only ever uses t_1 - t_0) can be expressed in one window.  Skipped intervals
 *         are run as dt = 0 (an exact no-op of the arithmetic, no divergence): their READINGS must be
 *         finite -- the separators the facades emit carry zeros.
 * first   [W] index of each window's first knot, or NULL for the dense layout knots[W][N+1][7].
 * count   [W] number of intervals of each window (<= N), or NULL = every window has N.  Values outside
 *         [0, N] are clamped into it by the kernels (the array lives in HBM and cannot be validated by the call).
 *         Windows may share knots (consecutive windows cut from one stream).
 * N       maximum number of intervals per window.
 * lin     [W][6]  {b_w_lin[3], b_a_lin[3]}  (CpiBase.h:113-114)
 * q_k_lin [W][4]  JPL q_GtoK linearisation orientation (CpiBase.h:115); required for model 2.
 * Zero-length windows produce the identity / zero state.  W == 0 is a no-op. */
int cpi_preintegrate_batch(cpi_ctx *ctx, const cpi_params *prm, int64_t W, int32_t N,
                           const double *knots, const int64_t *first, const int32_t *count,
                           const double *lin, const double *q_k_lin, const cpi_outputs *out);

/* Replaces: the whole caller side of GraphSolver::createimufactor_cpi_v1 / _v2 (GraphSolver_IMU.cpp:43-75, 97-130) for every
 * window of a trajectory at once, reading ONE IMU stream IN PLACE: no knot is copied, for any model and any output.
 *   stream        [K][7] knot records {t, w[3], a[3]} in arrival order, stamps NON-DECREASING (device memory)
 *   update_times  [U] non-decreasing: window u covers (update_times[u-1], update_times[u]] exactly as the reference cuts it
 *                 (GraphSolver_IMU.cpp:50-69): whole intervals while imu_times[1] <= updatetime, then the partial tail interval
 *                 with the front reading held, after which the front stamp is overwritten by the update time; window 0 starts
 *                 at the stream's first reading
 *   N             upper bound of the intervals of a window (whole + tail); a longer window is truncated to N intervals --
 *                 check cpi_stream_counts, which holds the TRUE counts after the call.  Give a TIGHT bound: the kernels'
 *                 loops run to the longest window of a wavefront, but the automatic lane split of the mean kernel
 *                 (cpi_params.lanes_per_window = 0) is chosen from N, not from the counts
 *   lin, q_k_lin  per window, as in cpi_preintegrate_batch
 *   workspace     cpi_stream_workspace_bytes(U) bytes of device memory, 16-byte aligned (28 bytes per window: where the
 *                 reference's deque stands at each update, found by interpolation search because the stamps are sorted).
 *                 After the call it holds the TRUE interval counts (cpi_stream_counts); the rest of its contents is
 *                 unspecified: a mean-only request (DT / alpha / beta / q and nothing else, models 1 and 2) runs NO cut kernel --
 *                 every wavefront of the mean kernel finds its own windows in its prologue and only the counts are written
 * The kernels patch the first knot's stamp and build the tail interval's closing knot from its predecessor in flight.
 * The readings must be finite: a NaN / Inf reading invalidates (only) the windows that contain it.
 * Results are bit-identical to cpi_preintegrate_batch on the knots / first / count that the host assemblers
 * (cpi_amd/stream.py, cpi_host::assemble_windows) produce from the same stream. */
size_t cpi_stream_workspace_bytes(int64_t U);
int cpi_preintegrate_stream(cpi_ctx *ctx, const cpi_params *prm, int64_t K, const double *stream, int64_t U,
                            const double *update_times, int32_t N, const double *lin, const double *q_k_lin,
                            void *workspace, const cpi_outputs *out);
const int32_t *cpi_stream_counts(const void *workspace, int64_t U);   /* device pointer into the workspace: count[U] */

/* The same loop for the mean outputs (DT, alpha, beta, q) on the TILED layout: the knots of 64 consecutive windows
 * interleaved per step,
 *     tiles[ceil(W/64)][N+1][7][64]      tiles[b][s][k][i] = field k of {t, w[3], a[3]} of knot s of window 64 b + i,
 * so that a wavefront (one tile, one lane per window) reads every step as seven coalesced 512-byte rows: one linear
 * stream per wavefront, no staging -- the mean-only recursion is HBM-bound and this is the layout it wants on MI355X
 * (DESIGN.md 3.1a).  Small batches (< 640 tiles) split a tile's steps over four wavefronts, same results to rounding
 * (prm->lanes_per_window = 1..8 pins the number of wavefronts per tile).
 * Columns past W inside the last tile are never written back.  count as in cpi_preintegrate_batch: a lane never reads
 * its column past row count[w], whatever lies there (unwritten memory, NaN) is harmless; rows of the tile array past the
 * largest count must still be ALLOCATED as the shape says.  Skipped intervals (dt <= 0, NaN dt) inside a window need
 * finite readings, as above.  Any Jacobian / covariance pointer in out -> CPI_ERR_INVALID (those kernels are FP64-bound:
 * the layout would buy nothing).
 *
 * PRODUCERS of the layout -- a caller never needs a dense copy first:
 *   cpi_assemble_tiles    cuts ONE IMU stream (device memory) into windows at successive update times and writes them
 *                         straight into tiles + count: the loop of GraphSolver::createimufactor_cpi_v1/v2
 *                         (GraphSolver_IMU.cpp:50-69 -- whole intervals while imu_times[1] <= updatetime, then the partial
 *                         tail interval with the front reading repeated, the front stamp overwritten by the update time)
 *                         for every window at once.  stream [K][7] knot records with NON-DECREASING stamps, update_times
 *                         [U] non-decreasing (then the deque state at the start of a window depends on the previous update
 *                         time alone; a stream with backward stamps needs the host assembler).  count[u] receives the TRUE
 *                         number of intervals of window u; rows beyond N are not written -- a caller sizes N >= max count
 *                         (the kernels clamp count to N).
 *                         WHEN: the copy costs 1.1-1.2 ms per 1 M x 50 windows (twice the tiled kernel it feeds), so a caller
 *                         that preintegrates a stream ONCE uses cpi_preintegrate_stream above (windows cut in place, no copy:
 *                         0.65-0.69 ms mean-only); tiles pay when the SAME windows are preintegrated again and again at new
 *                         linearisation points -- from about the 8th use (profiles/r05_assembler.md).
 *   cpi_tile_windows      re-tiles windows the caller already holds in the layouts of cpi_preintegrate_batch (dense
 *                         knots[W][N+1][7] with first == NULL, or a shared stream indexed by first[W] / count[W]); rows
 *                         past a window's last knot repeat that knot.  A full extra pass: for one-off use and tests.
 *                         cpi_tile_knots is the dense special case (first = count = NULL).
 *   host side             cpi_amd/csrc/cpi_host.hpp (assemble_windows_tiled, CpiBatch::flush_means) and
 *                         cpi_amd/stream.py (assemble_windows(..., layout="tiled")) write knot s of window w at
 *                         (((w / 64) (N+1) + s) 7 + k) 64 + w % 64 while they assemble; cpi_preintegrate_tiled_batch_host
 *                         takes such tiles from HOST memory through the chunked upload / kernel / download pipeline. */
int cpi_preintegrate_tiled_batch(cpi_ctx *ctx, const cpi_params *prm, int64_t W, int32_t N, const double *tiles,
                                 const int32_t *count, const double *lin, const double *q_k_lin, const cpi_outputs *out);
int cpi_assemble_tiles(cpi_ctx *ctx, int64_t K, const double *stream /*[K][7]*/, int64_t U, const double *update_times /*[U]*/,
                       int32_t N, double *tiles /*[ceil(U/64)][N+1][7][64]*/, int32_t *count /*[U]*/);
int cpi_tile_windows(cpi_ctx *ctx, int64_t W, int32_t N, const double *knots, const int64_t *first, const int32_t *count,
                     double *tiles);
int cpi_tile_knots(cpi_ctx *ctx, int64_t W, int32_t N, const double *knots /*[W][N+1][7]*/, double *tiles);

/* Replaces: ImuFactorCPIv1::evaluateError / ImuFactorCPIv2::evaluateError, one call per factor
 * (ImuFactorCPIv1.cpp:37-208, ImuFactorCPIv2.cpp:38-212), as driven by GTSAM's linearize loop.
 *
 * Factor f reads its measurement from the preintegration outputs of window f (field -> ctor
 * mapping of GraphSolver_IMU.cpp:74-75,129-130: J_b->J_beta, J_a->J_alpha, H_b->H_beta,
 * H_a->H_alpha, O_b->O_beta, O_a->O_alpha), its linearisation biases from lin[f] and, for
 * model 2, q_K_lin from q_k_lin[f].
 * states  [S][16] JPLNavState array; idx_i/idx_j [F] select state_i/state_j (NULL: f and f+1, which needs S >= F+1).
 *         S is the number of states: the device-pointer entries cannot inspect idx (it lives in HBM), so the kernels
 *         CLAMP every index into [0, S) -- a wrong index yields a wrong factor, never an out-of-bounds read; the
 *         _host variant validates the indices and returns CPI_ERR_INVALID.
 * err     [F][15] unwhitened residual; H1, H2 [F][225] dense column-major Jacobians wrt the two
 *         states' tangent vectors; H1/H2 may be NULL (the boost::optional<Matrix&> = none case). */
int cpi_factor_eval_batch(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                          const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                          const double *states, int64_t S, const int32_t *idx_i, const int32_t *idx_j,
                          double *err, double *H1, double *H2);

/* Fast path of the same evaluation for callers that assemble their own Hessian blocks (note on
 * a11): of the 450 doubles of the dense H1 / H2 pair only 54 depend on the current states; everything else is 0, +-I
 * or a copy of a measurement field the caller already holds.  packed [F][72] (576 B per factor), 3x3 blocks
 * column-major:
 *    [ 0..14]  err                                   (ImuFactorCPIv1.cpp:81-88)
 *    [15..23]  H1(0,0)   d e_theta / d theta_K       (:109)          [24..32]  H1(6,0)   d e_v / d theta_K   (:120)
 *    [33..41]  H1(12,0)  d e_p / d theta_K           (:132)          [42..50]  H1(0,3)   d e_theta / d b_g,K (:112)
 *    [51..59]  Rk = quat_2_Rot(q_GtoK)                               [60..68]  H2(0,0)   d e_theta / d theta_K+1 (:169)
 *    [69..71]  0 (padding to a multiple of 16 bytes)
 * The remaining blocks of the dense pair follow from these and the measurement (ImuFactorCPIv1.cpp:113-143,172-185):
 *    H1(3,3) = H1(9,9) = -I;  H1(6,3) = -J_beta;  H1(6,6) = -Rk;  H1(6,9) = -H_beta;  H1(12,3) = -J_alpha;
 *    H1(12,6) = -deltatime Rk;  H1(12,9) = -H_alpha;  H1(12,12) = -Rk;  H2 = blkdiag(H2(0,0), I, Rk, I, Rk);  all others 0.
 * Same arguments as cpi_factor_eval_batch.  (cpi_amd.unpack_factor in the Python mirror rebuilds the dense pair.) */
int cpi_factor_eval_packed_batch(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                                 const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                                 const double *states, int64_t S, const int32_t *idx_i, const int32_t *idx_j,
                                 double *packed);

/* Replaces: gtsam::noiseModel::Gaussian::Covariance(P_meas) in the factor constructors
 * (ImuFactorCPIv1.h:82, ImuFactorCPIv2.h:86).  GTSAM (bitbucket gtborg/gtsam @ c21186c6, not in the
 * reference tree) builds the square-root information  R = chol_upper(P^-1)  (Gaussian::Covariance ->
 * Gaussian::Information(cov.inverse()) -> Eigen::LLT::matrixU) once per factor, R^T R = P^-1.
 * Here R = B^-1 with P = B B^T, B upper triangular (the same matrix, obtained without forming P^-1).
 * P [F][225] column-major symmetric positive definite; sqrt_info [F][225] column-major upper triangular
 * (strict lower part written as zeros).  A non-positive pivot yields NaNs in that factor's R. */
int cpi_sqrt_information_batch(cpi_ctx *ctx, int64_t F, const double *P, double *sqrt_info);
/* The same factorisation on packed triangles (ABI 3): P_sym [F][120] in (cpi_outputs.P_sym of the covariance kernels), R_tri
 * [F][120] out -- 1 920 bytes per factor instead of 3 600, of which the dense form spends 840 on the mirrored half of P and
 * 840 on zeros it writes below R's diagonal.  Entry for entry the same values as cpi_sqrt_information_batch computes from the
 * dense form of the same P (same arithmetic, same order: bit-identical). */
int cpi_sqrt_information_packed_batch(cpi_ctx *ctx, int64_t F, const double *P_sym, double *R_tri);

/* cpi_factor_eval_batch followed by GTSAM's NoiseModelFactor::linearize whitening
 * (Gaussian::WhitenSystem): err <- R err, H1 <- R H1, H2 <- R H2 with R = sqrt_info[f]. */
int cpi_factor_eval_whitened_batch(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                                   const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                                   const double *states, int64_t S, const int32_t *idx_i, const int32_t *idx_j,
                                   const double *sqrt_info, double *err, double *H1, double *H2);
/* The same with R as its packed upper triangle R_tri [F][120] (cpi_sqrt_information_packed_batch): 840 bytes less to read per
 * factor, identical outputs (the 105 entries the dense form stores below the diagonal are zeros the kernel never multiplied by). */
int cpi_factor_eval_whitened_tri_batch(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                                       const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                                       const double *states, int64_t S, const int32_t *idx_i, const int32_t *idx_j,
                                       const double *R_tri, double *err, double *H1, double *H2);

/* The same evaluation carried one step further down GTSAM's pipeline: the factor's
 * contribution to the normal equations.  NoiseModelFactor::linearize turns (e, H1, H2) into the JacobianFactor
 * [A1 A2 | b] = [R H1, R H2 | -R e]; a HessianFactor built from it holds the augmented information matrix
 *     [A1 A2 b]^T [A1 A2 b] = [ G  g ; g^T  f ],   G = A^T A (30 x 30),  g = A^T b (30),  f = b^T b,
 * over the tangent vectors of (state_i, state_j).  hess [F][496]: its upper triangle, packed column-major
 * (entry (i, d), i <= d <= 30, at i + d (d + 1) / 2; rows / columns 0-14 state_i, 15-29 state_j, 30 the b column).
 * The whitened Jacobians never leave the chip.  (GTSAM is not in the reference tree: PARITY UNPINNED, checked against a
 * numpy restatement of the definition above on the outputs of cpi_factor_eval_whitened_batch.) */
int cpi_factor_hessian_batch(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                             const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                             const double *states, int64_t S, const int32_t *idx_i, const int32_t *idx_j,
                             const double *sqrt_info, double *hess);
/* ... with R as its packed upper triangle R_tri [F][120]: identical hess. */
int cpi_factor_hessian_tri_batch(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                                 const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                                 const double *states, int64_t S, const int32_t *idx_i, const int32_t *idx_j,
                                 const double *R_tri, double *hess);

/* Replaces: GraphSolver::getpredictedstate_v1 / _v2 (GraphSolver_IMU.cpp:263-281, 289-307):
 * states_j[f] = prediction of X(k+1) from states_i[idx_i[f]] and measurement f.  states_i [S][16]; idx_i NULL: state f. */
int cpi_predict_batch(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                      const cpi_outputs *meas, const double *states_i, int64_t S, const int32_t *idx_i,
                      double *states_j);

/* ---- Device sets: the 8-GPU path of a single-process host (nothing in the reference, which is a
 * single-threaded CPU program).  Windows (and factors) are independent units: rank r of n owns the contiguous block
 * [lo, hi) = cpi_shard_bounds(W, r, n) (block size ceil(W / n); trailing ranks may be short or empty), runs the ordinary
 * entries above on cpi_group_ctx(g, r) with pointers into ITS device's memory, and the one exchange step is
 * cpi_group_gather: every peer sends its output slab straight to the root (ncclSend / ncclRecv inside one
 * ncclGroupStart / End, rccl/rccl.h:700-722,923-933 -- each peer has its own xGMI link to the root, so a direct gather
 * is link-parallel where a ring would be per-link bound), the root's own block is a device-to-device copy.
 * cpi_group_create makes one context + one non-blocking HIP stream per device and, for n > 1, one RCCL communicator per
 * device (ncclCommInitAll, rccl.h:236; librccl.so.1 -- or the path in the environment variable CPI_AMD_RCCL_LIB, read at
 * that moment only -- is bound then, never before; failure to bind it is CPI_ERR_RCCL).  devices NULL = 0 .. n-1.
 * All calls are asynchronous on the group's streams; cpi_group_synchronize waits for every device.
 * Multi-process hosts (one rank per GPU) use torch.distributed instead: cpi_amd/dist.py issues the same pattern. */
typedef struct cpi_group cpi_group;
int cpi_group_create(int n, const int *devices, cpi_group **out);
void cpi_group_destroy(cpi_group *g);
int cpi_group_size(const cpi_group *g);
cpi_ctx *cpi_group_ctx(cpi_group *g, int rank);                 /* borrowed; owned by the group */
const char *cpi_group_last_error(const cpi_group *g);           /* g may be NULL: last error of a failed create */
void cpi_shard_bounds(int64_t W, int rank, int n, int64_t *lo, int64_t *hi);
/* local[r] = the outputs of rank r's block (device pointers on device r, hi - lo windows each); root_out = arrays of W
 * windows on the root's device: rank r's block lands at window offset lo.  Every field that is non-NULL in root_out
 * must be non-NULL in every non-empty local[r].  local[root] may already point into root_out (no copy then).
 * When every peer's outputs are ONE SLAB (the wanted fields back to back, field-major over Wb >= hi - lo windows: what
 * cpi_outputs_bind_slab lays out) the exchange is ONE message per peer into a staging area on the root + one unpack
 * launch there; separately allocated fields cost one message per (peer, field), received in place.
 * cpi_group_last_gather_messages: messages per peer of the last gather (1 = slab path). */
int cpi_group_gather(cpi_group *g, int root, int64_t W, const cpi_outputs *local, const cpi_outputs *root_out);
int cpi_group_last_gather_messages(const cpi_group *g);
/* The exchange INSIDE one batch (ABI 3; DESIGN.md section 7 has the wire budget that asks for it: at configs[4]'s full-V1 outputs a
 * peer's slab is 1.4 - 2.2 GB over ONE xGMI link (18 - 29 ms at 76.8 GB/s one way), about as long as the 26 ms of kernels that produce it -- issued after them, all of it is
 * exposed).  Every rank's block is cut into `chunks` sub-blocks of cper = ceil(ceil(W / n) / chunks) windows
 * (cpi_shard_chunk_bounds: sub-block c of rank r = [lo_r + c cper, min(hi_r, lo_r + (c + 1) cper)); trailing ones may be short or
 * empty).  The host enqueues the ordinary entries for sub-block c on cpi_group_ctx(g, r) and then calls
 * cpi_group_gather_chunk(g, root, W, c, chunks, local_c, root_out): sub-block c of every rank travels to the root on the group's
 * EXCHANGE streams (a second non-blocking stream per device, made at the first use), behind everything that is enqueued on the
 * ranks' compute streams at the moment of the call -- and the compute streams stay free for sub-block c + 1, whose kernels
 * run while sub-block c is on the wire.  local_c[r] = the outputs of sub-block c of rank r, a slab of its own
 * (cpi_outputs_bind_slab over >= the sub-block's windows: ONE message per peer and chunk) or separately allocated fields;
 * root_out = arrays of W windows, as for cpi_group_gather.  The call with chunk == chunks - 1 JOINS: every rank's compute stream
 * waits for its exchange stream, so whatever is enqueued later on the contexts -- and cpi_group_synchronize, which also waits
 * for the exchange streams -- is ordered behind the whole exchange.  Buffers of sub-block c must not be rewritten before that
 * join (or a cpi_group_synchronize).  chunks == 1: cpi_group_gather, issued on the exchange streams. */
void cpi_shard_chunk_bounds(int64_t W, int rank, int n, int chunk, int chunks, int64_t *lo, int64_t *hi);
int cpi_group_gather_chunk(cpi_group *g, int root, int64_t W, int chunk, int chunks, const cpi_outputs *local_chunk,
                           const cpi_outputs *root_out);
/* Slab layout of an output set: the fields that are non-NULL in `mask`, back to back in the order of cpi_outputs (P_sym last:
 * a mask with P_sym instead of P makes the slab -- and the message a peer sends -- 840 bytes per window shorter), each over
 * Wb windows.  _slab_doubles: size of the slab; _bind_slab: *bound = mask's fields pointing into slab (others NULL). */
size_t cpi_outputs_slab_doubles(const cpi_outputs *mask, int64_t Wb);
int cpi_outputs_bind_slab(const cpi_outputs *mask, int64_t Wb, double *slab, cpi_outputs *bound);
int cpi_group_synchronize(cpi_group *g);

/* For host-side callers (the CpiV1-shaped C++ facade in cpi_amd/csrc/cpi_host.hpp): same as cpi_preintegrate_batch
 * but every pointer is a HOST pointer; stages through device memory owned by the context and returns when the outputs
 * are in host memory.  Dense batches (first == NULL) run as an upload / kernels / download pipeline over chunks of
 * <= 65536 windows: with PINNED host buffers (cpi_host_alloc, hipHostMalloc) the three overlap (PCIe is full duplex);
 * with pageable memory the result is the same, the copies serialise.  Lanes per window are chosen per chunk, so the
 * rounding of a window may differ from the device-pointer call on the whole batch (set prm->lanes_per_window to pin
 * it).  PCIe-inclusive, never the benchmarked path. */
int cpi_preintegrate_batch_host(cpi_ctx *ctx, const cpi_params *prm, int64_t W, int32_t N,
                                const double *knots, const int64_t *first, const int32_t *count,
                                int64_t n_knots, const double *lin, const double *q_k_lin,
                                const cpi_outputs *out);
/* The tiled layout from HOST memory (tiles written by a host-side assembler), mean outputs only: same pipeline, chunks of
 * 1024 tiles. */
int cpi_preintegrate_tiled_batch_host(cpi_ctx *ctx, const cpi_params *prm, int64_t W, int32_t N, const double *tiles,
                                      const int32_t *count, const double *lin, const double *q_k_lin, const cpi_outputs *out);
/* page-locked host memory for the entries above (hipHostMalloc / hipHostFree); NULL when the allocation fails */
/* cpi_preintegrate_stream with HOST pointers -- what a caller shaped like GraphSolver::createimufactor_cpi_v1/v2
 * (GraphSolver_IMU.cpp:34-134) holds: its IMU deque as one array stream[K][7], the update times of the states it creates,
 * one linearisation point per window; the measurements of every window come back in host memory.  count (may be NULL)
 * receives the TRUE interval count of every window (> N: that window was truncated to N intervals).  The stream is uploaded
 * once, whole; the windows are cut on the device.  PCIe-inclusive, never the benchmarked path. */
int cpi_preintegrate_stream_host(cpi_ctx *ctx, const cpi_params *prm, int64_t K, const double *stream, int64_t U,
                                 const double *update_times, int32_t N, const double *lin, const double *q_k_lin,
                                 const cpi_outputs *out, int32_t *count);
void *cpi_host_alloc(size_t bytes);
void cpi_host_free(void *p);
int cpi_factor_eval_batch_host(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                               const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                               const double *states, int64_t S, const int32_t *idx_i,
                               const int32_t *idx_j, double *err, double *H1, double *H2);

#ifdef __cplusplus
}
#endif
#endif /* CPI_AMD_H */
