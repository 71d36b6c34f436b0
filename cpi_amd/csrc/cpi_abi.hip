// cpi_abi.hip -- the C-ABI of include/cpi_amd.h: argument checks, launch heuristics, device sets (RCCL over xGMI) and the
// host-pointer pipelines.  Host code only; the kernels live in cpi_mean.hip / cpi_cov.hip / cpi_factor.hip and are reached
// through cpi::launch (cpi_args.hpp).
//
// Kernels behind the entries (one 64-lane wavefront per workgroup; all arithmetic f64 VALU, no MFMA -- the contractions
// are 3x3 / sparse 15x15):
//   cpi_mean_kernel<MODEL,JAC,AVG,L>   means (+ analytic bias Jacobians), L lanes per window.  CpiV1.h:67-259 / CpiV2.h:88-305.
//   cpi_mean_tiled_kernel              the same recursion on the tiled input layout (one lane per window, no staging).
//   cpi_cov_kernel<MODEL,AVG>          covariance (model 2: + compounded state transition -> Jacobians) and means;
//                                      column-lane RK4 recursion.  CpiV1.h:266-353 / CpiV2.h:314-464.
//   cpi_forster_kernel                 GTSAM's discrete comparator.  GraphSolver_IMU.cpp:141-232.
//   cpi_factor_kernel<MODEL,WHITEN,LPF,TRI> / cpi_factor_packed_kernel / cpi_factor_hessian_kernel<MODEL,TRI>
//                                      evaluateError residual + Jacobian blocks.  ImuFactorCPIv1.cpp:37-208 / v2.cpp:38-212.
//                                      TRI: the square-root information arrives as its packed upper triangle (ABI 3).
//   cpi_sqrt_info_kernel<PACKED>       R = chol_upper(P^-1) per factor (ImuFactorCPIv1.h:82); PACKED: triangles in and out.
//   cpi_predict_kernel<MODEL>          GraphSolver_IMU.cpp:263-307.
// The covariance / Forster kernels write cpi_outputs.P (dense) and / or P_sym (packed upper triangle, ABI 3).
//   cpi_tile_knots_kernel / cpi_assemble_tiles_kernel   producers of the tiled layout (GraphSolver_IMU.cpp:50-69).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // prototypes and enums only: the library is bound lazily with dlopen (never linked)
#include <dlfcn.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/cpi_amd.h"
#ifdef CPI_TEST_HOOKS
#include "../../include/cpi_amd_test.h"   // the two test hooks exist in libcpi_amd_test.so only (python -m cpi_amd.build --test-hooks)
#endif
#include "cpi_args.hpp"

using namespace cpi;

constexpr int kOutFields = 13;   // fields of cpi_outputs (ABI 3: P_sym is the 13th)
static_assert(sizeof(cpi_outputs) == kOutFields * sizeof(double *), "cpi_outputs is a plain table of kOutFields pointers");

// ============================================================================================
// contexts
// ============================================================================================
struct HostPipe;
static void host_pipe_destroy(HostPipe *);
struct cpi_ctx {
    int device;
    hipStream_t stream;
    std::string err;
    // side stream + fork / join events (created at the first use): the two INDEPENDENT kernels of a "model 1, everything"
    // request (covariance kernel; analytic-Jacobian kernel) run concurrently -- see cpi_preintegrate_batch
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    unsigned big_lds_set = 0;   // bit per kernel instantiation whose dynamic-LDS limit was raised on this device
    struct HostPipe *pipe = nullptr;   // staging of the host-pointer entries (created at their first use)
};
static thread_local std::string g_create_err;

static int fail(cpi_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg; else g_create_err = msg;
    return code;
}
// Selects the context's device for the duration of a call and restores the caller's current device afterwards.
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    hipError_t enter(int dev) {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return e;
        if (prev == dev) return hipSuccess;
        e = hipSetDevice(dev);
        changed = (e == hipSuccess);
        return e;
    }
    ~DeviceGuard() { if (changed) (void)hipSetDevice(prev); }
};
#define CPI_HIP(ctx, call)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, CPI_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));   \
    } while (0)

extern "C" int cpi_abi_version(void) { return CPI_ABI_VERSION; }
#ifndef CPI_BUILD_ID
#define CPI_BUILD_ID "unknown"
#endif
// "cpi-build-id:<id>" as ONE string in .rodata: cpi_amd/_lib.py finds the id of a library file by that tag without loading it
static const char kBuildIdTagged[] = "cpi-build-id:" CPI_BUILD_ID;
extern "C" const char *cpi_build_id(void) { return kBuildIdTagged + 13; }

extern "C" int cpi_ctx_create(int device, void *stream, cpi_ctx **out) {
    if (!out) return fail(nullptr, CPI_ERR_INVALID, "cpi_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, CPI_ERR_NO_DEVICE, "cpi_ctx_create: no HIP device available (this library has no CPU fallback)");
    if (device < 0) {
        e = hipGetDevice(&device);
        if (e != hipSuccess) return fail(nullptr, CPI_ERR_HIP, std::string("hipGetDevice: ") + hipGetErrorString(e));
    }
    if (device >= ndev) return fail(nullptr, CPI_ERR_INVALID, "cpi_ctx_create: device index out of range");
    cpi_ctx *c = new cpi_ctx();
    c->device = device;
    c->stream = (hipStream_t)stream;
    *out = c;
    return CPI_OK;
}
extern "C" void cpi_ctx_destroy(cpi_ctx *ctx) {
    if (!ctx) return;
    if (ctx->side || ctx->pipe) {
        int prev = -1;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(ctx->device);
        if (ctx->side) {
            (void)hipStreamDestroy(ctx->side);
            (void)hipEventDestroy(ctx->ev_fork);
            (void)hipEventDestroy(ctx->ev_join);
        }
        if (ctx->pipe) host_pipe_destroy(ctx->pipe);
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    delete ctx;
}
extern "C" int cpi_ctx_set_stream(cpi_ctx *ctx, void *stream) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    ctx->stream = (hipStream_t)stream;
    return CPI_OK;
}
extern "C" const char *cpi_last_error(const cpi_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }
extern "C" int cpi_ctx_synchronize(cpi_ctx *ctx) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    CPI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CPI_OK;
}

// hipLaunchKernelGGL takes 32-bit grid dimensions: a launch is refused rather than silently truncated
static bool grid_ok(long long nb) { return nb > 0 && nb <= 0x7fffffffLL; }

#ifdef CPI_EXPERIMENTS
// Measurement switches of tools/exp/ (A/B runs of kernel variants).  They exist ONLY in a -DCPI_EXPERIMENTS build
// (python -m cpi_amd.build --experiments -> libcpi_amd_exp.so, loaded through CPI_AMD_LIB); the default library reads no
// environment variable on any launch path.  Each is read once per process.
namespace expsw {
static launch::MeanDmaCfg mean_dma() {   // CPI_AMD_MEAN_DMA = "off" | "KC,S,A" (A = 1: 16-byte aligned pieces)
    static launch::MeanDmaCfg c = [] {
        launch::MeanDmaCfg d = {0, 0, 0};
        const char *e = getenv("CPI_AMD_MEAN_DMA");
        if (e && strcmp(e, "off") != 0) { int k = 0, s = 0, al = 1; if (sscanf(e, "%d,%d,%d", &k, &s, &al) >= 2) d = {k, s, al}; }
        return d;
    }();
    return c;
}
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
static int mean_blk() { static int v = [] { const char *e = getenv("CPI_AMD_MEAN_BLK"); return e ? (strcmp(e, "off") == 0 ? -1 : atoi(e)) : 0; }(); return v; }
static bool mean_line() { static int v = env_int("CPI_AMD_MEAN_LINE", 0); return v != 0; }   // cpi_mean_line_kernel on the leading groups of a dense one-lane batch
static int blk_mode() { static int v = env_int("CPI_AMD_BLK_MODE", 0); return v; }
static int probe_lds() { static int v = env_int("CPI_AMD_PROBE_LDS", 0); return v; }
static bool no_overlap() { static int v = env_int("CPI_AMD_NO_OVERLAP", 0); return v != 0; }
static bool no_fused_cut() { static int v = env_int("CPI_AMD_NO_FUSED_CUT", 0); return v != 0; }   // A/B: the workspace route for mean-only streams
static int factor_lanes() { static int v = env_int("CPI_AMD_FACTOR_LANES", 0); return (v == 16 || v == 8 || v == 4) ? v : 0; }
static int packed_lpf() { static int v = env_int("CPI_AMD_PACKED_LPF", 0); return (v == 2 || v == 3 || v == 4 || v == 6 || v == 8) ? v : 0; }
}  // namespace expsw
#endif

// ============================================================================================
// preintegration
// ============================================================================================
static int pick_lanes(const cpi_params *prm, int64_t W, int N, bool jac) {
    // model 2 with analytic Jacobians: sequential per window (the O_a / O_b recursion is not composed); model 2
    // mean-only composes through the gravity response matrices at roughly twice the arithmetic per interval
    if (prm->model == CPI_MODEL_V2 && jac) return 1;
    // (model 2's level cost re-fitted in round 6 on windows of 10 and 20 intervals -- 10 k windows: 4 lanes 8.8 / 10.9 us, the 5 the old
    //  0.6 picked 9.3 / 11.5; the choices at N = 50 do not move.  Model 1's automatic choice is within 0-3 % of the best lane count for
    //  N = 10 / 20 at 5 k ... 50 k windows: profiles/r06_short_windows.md)
    const double t_int = (prm->model == CPI_MODEL_V2) ? 1.2 : 0.55, t_lvl = (prm->model == CPI_MODEL_V2) ? 1.3 : 0.3;
    int L = prm->lanes_per_window;
    if (L <= 0) {
        // Small batches are latency-bound: as long as every wavefront gets a SIMD of its own (<= 1024 wavefronts
        // on MI355X) the launch lasts as long as one wavefront -- intervals per lane plus composition levels
        // (measured: ~0.55 us per interval, ~0.3 us per level).  Splitting further makes wavefronts share SIMDs
        // and loses; batches with more than 1024 single-lane wavefronts are throughput-bound and want L = 1.
        // Measured optima: L = 12 at 5 k windows x 50, 6 at 10 k, 4 at 15 k, 3 at 20 k, 2 at 30 k, 1 from ~60 k.
        double best = 1e300;
        L = 1;
        const int *choices = nullptr;
        const int nc = launch::mean_lane_choices(&choices);
        for (int i = 0; i < nc; i++) {
            const int c = choices[i];
            if (c > 1 && 2 * c > N) break;
            const int64_t waves = (W + (64 / c) - 1) / (64 / c);
            if (c > 1 && waves > 1024) break;
            int levels = 0;
            while ((1 << levels) < c) levels++;
            const double cost = t_int * (double)((N + c - 1) / c) + t_lvl * levels;
            if (cost < best) { best = cost; L = c; }
        }
    }
    return L;
}

#ifdef CPI_EXPERIMENTS
static PreArgs shift_windows(const PreArgs &a, long long w0) {
    PreArgs t = a;
    t.W = a.W - w0;
    if (a.first) t.first = a.first + w0; else t.knots = a.knots + w0 * (long long)(a.N + 1) * 7;
    if (a.count) t.count = a.count + w0;
    t.lin = a.lin + w0 * 6;
    if (a.qk) t.qk = a.qk + w0 * 4;
    static const int n[kOutFields] = { 1, 3, 3, 4, 9, 9, 9, 9, 9, 9, 9, 225, 120 };
    double **f[kOutFields] = { &t.out.DT, &t.out.alpha, &t.out.beta, &t.out.q, &t.out.J_q, &t.out.J_a, &t.out.J_b, &t.out.H_a,
                               &t.out.H_b, &t.out.O_a, &t.out.O_b, &t.out.P, &t.out.P_sym };
    for (int k = 0; k < kOutFields; k++) if (*f[k]) *f[k] += w0 * n[k];
    return t;
}
#endif

// cpi_preintegrate_stream: knots = ONE stream of K readings cut at update[W]; the workspace arrays are filled by
// cpi_cut_windows_kernel when a kernel that reads them runs (covariance / Forster / analytic-Jacobian kernels), and stay
// untouched -- except count -- when the mean kernel cuts its own windows (mean-only requests)
struct StreamCut {
    long long K;
    const double *update;
    long long *first;
    int *count;
    double *tstart, *tend;
};
static int preintegrate_impl(cpi_ctx *ctx, const cpi_params *prm, int64_t W, int32_t N, const double *knots, const int64_t *first,
                             const int32_t *count, const StreamCut *sc, const double *lin,
                             const double *q_k_lin, const cpi_outputs *out);
extern "C" int cpi_preintegrate_batch(cpi_ctx *ctx, const cpi_params *prm, int64_t W, int32_t N,
                                      const double *knots, const int64_t *first, const int32_t *count,
                                      const double *lin, const double *q_k_lin, const cpi_outputs *out) {
    return preintegrate_impl(ctx, prm, W, N, knots, first, count, nullptr, lin, q_k_lin, out);
}
static int preintegrate_impl(cpi_ctx *ctx, const cpi_params *prm, int64_t W, int32_t N, const double *knots, const int64_t *first,
                             const int32_t *count, const StreamCut *sc, const double *lin,
                             const double *q_k_lin, const cpi_outputs *out) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (!prm || !out) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_batch: prm/out is NULL");
    if (prm->model != CPI_MODEL_V1 && prm->model != CPI_MODEL_V2 && prm->model != CPI_MODEL_FORSTER)
        return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_batch: model must be 1, 2 or 3 (CPI_MODEL_FORSTER)");
    if (W < 0 || N < 0) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_batch: negative size");
    if (W == 0) return CPI_OK;
    if (!knots || !lin) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_batch: knots/lin is NULL");
    if (prm->model == CPI_MODEL_V2 && !q_k_lin)
        return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_batch: model 2 needs q_k_lin");
    if (!grid_ok(W)) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_batch: W exceeds 2^31 - 1 windows per call (32-bit grid)");
    if (N > 65535) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_batch: N (intervals per window) must be <= 65535");
    int L = prm->lanes_per_window;
    if (L != 0 && !launch::mean_lanes_supported(L)) return fail(ctx, CPI_ERR_INVALID, "lanes_per_window must be 0 or one of 1,2,3,4,5,6,8,12,16,32,64");

    const bool want_mean = out->DT || out->alpha || out->beta || out->q;
    const bool want_jac = out->J_q || out->J_a || out->J_b || out->H_a || out->H_b || out->O_a || out->O_b;
    const bool want_cov = out->P != nullptr || out->P_sym != nullptr;
    const bool forster = prm->model == CPI_MODEL_FORSTER;
    const bool avg = prm->imu_avg != 0;
    const bool v2 = prm->model == CPI_MODEL_V2;
    const bool stj = v2 && prm->state_transition_jacobians != 0;
    // Which kernel owns what:
    //   covariance kernel : P, and (model 2 + state_transition_jacobians) the Jacobians read out of
    //                       Discrete_J_b; it also carries the means, so it writes them when it runs.
    //   mean kernel       : means when no covariance kernel runs; the ANALYTIC Jacobians (model 1 always,
    //                       model 2 when state_transition_jacobians == 0).
    const bool anything = want_mean || want_jac || want_cov;
    const bool run_cov = !forster && (want_cov || (stj && want_jac));
    const bool mean_jac = !forster && want_jac && !stj;
    const bool run_mean = !forster && (mean_jac || (want_mean && !run_cov));
    // A stream: the mean-only kernel cuts its own windows (fused; it still leaves the TRUE counts in the workspace); every
    // other kernel reads the cut that cpi_cut_windows_kernel leaves there.  Nothing asked for: the counts are still owed.
    bool fused_cut = sc && anything && run_mean && !mean_jac && !run_cov && sc->K >= 4;
#ifdef CPI_EXPERIMENTS
    if (expsw::no_fused_cut()) fused_cut = false;
#endif

    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    if (sc && !fused_cut) {
        launch::cut_windows(sc->K, knots, (long long)W, sc->update, (int)N, sc->first, sc->count, sc->tstart, sc->tend, ctx->stream);
        CPI_HIP(ctx, hipGetLastError());
        first = reinterpret_cast<const int64_t *>(sc->first); count = sc->count;
    }
    if (!anything) return CPI_OK;
    PreArgs a;
    memset(&a, 0, sizeof a);
    a.W = W; a.N = N; a.knots = knots; a.first = (const long long *)first; a.count = count;
    a.lin = lin; a.qk = q_k_lin;
    if (sc) { a.K = sc->K; if (!fused_cut) { a.tstart = sc->tstart; a.tend = sc->tend; } }
    for (int i = 0; i < 3; i++) a.grav[i] = prm->grav[i];
    a.q4[0] = prm->sigma_w * prm->sigma_w; a.q4[1] = prm->sigma_wb * prm->sigma_wb;
    a.q4[2] = prm->sigma_a * prm->sigma_a; a.q4[3] = prm->sigma_ab * prm->sigma_ab;
    a.out = *out;
    if (forster) {   // one kernel owns everything; imu_avg, q_k_lin, grav play no part
        launch::forster(a, ctx->stream);
        CPI_HIP(ctx, hipGetLastError());
        return CPI_OK;
    }

    // Model 1 with Jacobians AND covariance is two independent kernels over the same knots (disjoint outputs).  The
    // covariance kernel waits on the LDS pipe about as much as it issues VALU work and uses no more than two wavefronts per
    // SIMD; the Jacobian kernel is pure FP64 VALU with no LDS: issued on a side stream (fork / join by events, so everything
    // later on the context's stream still waits for both, and a stream capture sees an ordinary fork) they share the SIMDs.
    bool overlap_on = true;
#ifdef CPI_EXPERIMENTS
    overlap_on = !expsw::no_overlap();
#endif
    hipStream_t mean_stream = ctx->stream;
    bool forked = false;
    if (run_cov && run_mean && mean_jac && overlap_on) {
        if (!ctx->side) {
            CPI_HIP(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
            CPI_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
            CPI_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        }
        CPI_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        CPI_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
        mean_stream = ctx->side;
        forked = true;
    }
    if (run_cov) {
        PreArgs c = a;
        c.write_means = want_mean ? 1 : 0;
        c.write_jac = (stj && want_jac) ? 1 : 0;
        launch::cov(prm->model, avg, c, ctx->stream);
    }
    if (run_mean) {
        PreArgs m = a;
        m.write_means = (want_mean && !run_cov) ? 1 : 0;
        m.write_jac = mean_jac ? 1 : 0;
        if (fused_cut) { m.update = sc->update; m.count_out = sc->count; m.first = nullptr; m.count = nullptr; }
        const int LL = pick_lanes(prm, W, N, mean_jac);
        long long done = 0;
#ifdef CPI_EXPERIMENTS
        const int bl = expsw::mean_blk();
        m.dbg = expsw::blk_mode();
        // (both experimental kernels address `knots` as the dense [W][N + 1][7] layout: never on a stream call, whose outer
        //  first / count are NULL on the fused-cut route although the windows are anything but dense)
        if (!sc && !mean_jac && !first && bl > 0 && (size_t)(64 / bl) * (size_t)(N + 1) * 56 <= 65536 && N >= 1) {
            if (launch::mean_blk(prm->model, bl, avg, m, ctx->stream)) done = W;
        }
        const launch::MeanDmaCfg dc = expsw::mean_dma();
        if (!sc && !done && !mean_jac && LL == 1 && !first && !count && dc.kc > 0 && N >= 2 * dc.kc)
            done = launch::mean_dma(prm->model, dc, avg, m, ctx->stream);
        if (!sc && !done && !mean_jac && LL == 1 && !first && !count && expsw::mean_line()) done = launch::mean_line(prm->model, avg, m, ctx->stream);
        if (done && done < W) m = shift_windows(m, done);
#endif
        if (done < W) launch::mean(prm->model, mean_jac, avg, LL, m, mean_stream);
    }
    if (forked) {
        CPI_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->side));
        CPI_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    }
    CPI_HIP(ctx, hipGetLastError());
    return CPI_OK;
}

// Replaces the caller-side loop of GraphSolver::createimufactor_cpi_v1 / _v2 (GraphSolver_IMU.cpp:43-75, 97-130) for ALL the
// windows of a trajectory at once, with ZERO copies of the IMU data: cpi_cut_windows_kernel finds, per update time, where the
// reference's deque would stand (28 bytes per window into the caller's workspace), and the preintegration kernels read the
// stream in place, patching the first knot's stamp and building a partial tail interval from its predecessor in flight.
static size_t align16(size_t n) { return (n + 15) & ~(size_t)15; }
extern "C" size_t cpi_stream_workspace_bytes(int64_t U) {
    if (U <= 0) return 16;
    return align16((size_t)U * 8) * 3 + align16((size_t)U * 4);
}
extern "C" const int32_t *cpi_stream_counts(const void *workspace, int64_t U) {
    if (!workspace || U <= 0) return nullptr;
    return reinterpret_cast<const int32_t *>(static_cast<const char *>(workspace) + align16((size_t)U * 8) * 3);
}
extern "C" int cpi_preintegrate_stream(cpi_ctx *ctx, const cpi_params *prm, int64_t K, const double *stream, int64_t U,
                                       const double *update_times, int32_t N, const double *lin, const double *q_k_lin,
                                       void *workspace, const cpi_outputs *out) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (K < 0 || U < 0 || N < 0) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_stream: negative size");
    if (U == 0) return CPI_OK;
    if (K == 0) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_stream: the stream is empty");
    if (!stream || !update_times || !workspace) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_stream: NULL argument");
    if (((uintptr_t)workspace & 15) != 0) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_stream: the workspace must be 16-byte aligned");
    if (!grid_ok(U)) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_stream: U exceeds 2^31 - 1 windows per call");
    // everything preintegrate_impl would refuse is refused HERE, before the cut kernel is enqueued: an invalid call must not
    // leave a launch behind that writes the caller's workspace
    if (!prm || !out || !lin) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_stream: prm/out/lin is NULL");
    if (prm->model != CPI_MODEL_V1 && prm->model != CPI_MODEL_V2 && prm->model != CPI_MODEL_FORSTER)
        return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_stream: model must be 1, 2 or 3 (CPI_MODEL_FORSTER)");
    if (prm->model == CPI_MODEL_V2 && !q_k_lin) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_stream: model 2 needs q_k_lin");
    if (N > 65535) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_stream: N (intervals per window) must be <= 65535");
    if (prm->lanes_per_window != 0 && !launch::mean_lanes_supported(prm->lanes_per_window))
        return fail(ctx, CPI_ERR_INVALID, "lanes_per_window must be 0 or one of 1,2,3,4,5,6,8,12,16,32,64");
    char *ws = static_cast<char *>(workspace);
    StreamCut sc;
    sc.K = (long long)K; sc.update = update_times;
    sc.first = reinterpret_cast<long long *>(ws);
    sc.tstart = reinterpret_cast<double *>(ws + align16((size_t)U * 8));
    sc.tend = reinterpret_cast<double *>(ws + 2 * align16((size_t)U * 8));
    sc.count = reinterpret_cast<int *>(ws + 3 * align16((size_t)U * 8));
    return preintegrate_impl(ctx, prm, U, N, stream, nullptr, nullptr, &sc, lin, q_k_lin, out);
}

// ============================================================================================
// re-linearisation sweeps
// ============================================================================================
// Lanes per factor of the dense sweep: 16 gives the most wavefronts (small sweeps), fewer lanes do less redundant
// arithmetic.  Measured on MI355X, 1 M factors: plain 0.82 ms with 8 lanes vs 1.09 ms with 16; whitened (37 KB vs 19 KB of
// LDS per wavefront) 1.45 ms with 8 vs 1.37 ms with 16.
static int factor_lanes(int64_t F, bool whiten) {
#ifdef CPI_EXPERIMENTS
    if (expsw::factor_lanes()) return expsw::factor_lanes();
#endif
    if (whiten) return 16;
    return F >= 32768 ? 8 : 16;
}

// argument checks shared by the four sweeps; fills the kernel argument block
static int factor_args(cpi_ctx *ctx, const char *who, int32_t model, const double grav[3], int64_t F, const cpi_outputs *meas,
                       const double *lin, const double *q_k_lin, const double *states, int64_t S, const int32_t *idx_i,
                       const int32_t *idx_j, FactorArgs &a) {
    const std::string w(who);
    if (model != CPI_MODEL_V1 && model != CPI_MODEL_V2) return fail(ctx, CPI_ERR_INVALID, w + ": model must be 1 or 2");
    if (F < 0) return fail(ctx, CPI_ERR_INVALID, w + ": negative size");
    if (!grav || !meas || !lin || !states) return fail(ctx, CPI_ERR_INVALID, w + ": NULL argument");
    if (!meas->DT || !meas->alpha || !meas->beta || !meas->q || !meas->J_q || !meas->J_a || !meas->J_b || !meas->H_a || !meas->H_b)
        return fail(ctx, CPI_ERR_INVALID, w + ": measurement fields DT/alpha/beta/q/J_q/J_a/J_b/H_a/H_b are required");
    if (model == CPI_MODEL_V2 && (!q_k_lin || !meas->O_a || !meas->O_b))
        return fail(ctx, CPI_ERR_INVALID, w + ": model 2 needs q_k_lin, O_a, O_b");
    if (!grid_ok(F)) return fail(ctx, CPI_ERR_INVALID, w + ": F exceeds 2^31 - 1 factors per call");
    if (S <= 0 || (!idx_i && S < F) || (!idx_j && S < F + 1))
        return fail(ctx, CPI_ERR_INVALID, w + ": S (number of states) must be >= 1, and >= F + 1 when idx_i / idx_j are NULL (chained states f, f + 1)");
    memset(&a, 0, sizeof a);
    a.F = F;
    for (int i = 0; i < 3; i++) a.grav[i] = grav[i];
    a.meas = *meas; a.lin = lin; a.qk = q_k_lin; a.states = states; a.S = S; a.idx_i = idx_i; a.idx_j = idx_j;
    return CPI_OK;
}

static int factor_eval_impl(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F, const cpi_outputs *meas,
                            const double *lin, const double *q_k_lin, const double *states, int64_t S, const int32_t *idx_i,
                            const int32_t *idx_j, const double *sqrt_info, bool tri, double *err, double *H1, double *H2) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (F == 0) return CPI_OK;
    if (!err) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_batch: NULL argument");
    FactorArgs a;
    const int rc = factor_args(ctx, "cpi_factor_eval_batch", model, grav, F, meas, lin, q_k_lin, states, S, idx_i, idx_j, a);
    if (rc != CPI_OK) return rc;
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    a.err = err; a.H1 = H1; a.H2 = H2; a.sqrt_info = sqrt_info; a.r_tri = tri ? 1 : 0;
    launch::factor(model, sqrt_info != nullptr, factor_lanes(F, sqrt_info != nullptr), a, ctx->stream);
    CPI_HIP(ctx, hipGetLastError());
    return CPI_OK;
}

extern "C" int cpi_factor_eval_batch(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                                     const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                                     const double *states, int64_t S, const int32_t *idx_i, const int32_t *idx_j,
                                     double *err, double *H1, double *H2) {
    return factor_eval_impl(ctx, model, grav, F, meas, lin, q_k_lin, states, S, idx_i, idx_j, nullptr, false, err, H1, H2);
}

extern "C" int cpi_factor_eval_whitened_batch(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                                              const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                                              const double *states, int64_t S, const int32_t *idx_i, const int32_t *idx_j,
                                              const double *sqrt_info, double *err, double *H1, double *H2) {
    if (ctx && !sqrt_info) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_whitened_batch: sqrt_info is NULL");
    return factor_eval_impl(ctx, model, grav, F, meas, lin, q_k_lin, states, S, idx_i, idx_j, sqrt_info, false, err, H1, H2);
}
extern "C" int cpi_factor_eval_whitened_tri_batch(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                                                  const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                                                  const double *states, int64_t S, const int32_t *idx_i, const int32_t *idx_j,
                                                  const double *R_tri, double *err, double *H1, double *H2) {
    if (ctx && !R_tri) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_whitened_tri_batch: R_tri is NULL");
    return factor_eval_impl(ctx, model, grav, F, meas, lin, q_k_lin, states, S, idx_i, idx_j, R_tri, true, err, H1, H2);
}

extern "C" int cpi_factor_eval_packed_batch(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                                            const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                                            const double *states, int64_t S, const int32_t *idx_i, const int32_t *idx_j,
                                            double *packed) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (F == 0) return CPI_OK;
    if (!packed) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_packed_batch: NULL argument");
    FactorArgs a;
    const int rc = factor_args(ctx, "cpi_factor_eval_packed_batch", model, grav, F, meas, lin, q_k_lin, states, S, idx_i, idx_j, a);
    if (rc != CPI_OK) return rc;
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    // lanes per factor.  Every lane of a factor repeats the shared quaternion algebra, so fewer lanes = less VALU per factor
    // but more LDS per wavefront (a factor's staged record + packed output = 1.5 KB).  Measured (MI355X, 1 M factors, model 1 /
    // model 2, us): 8 lanes 383 / 435 (VALU 53 % busy at 2 wavefronts per SIMD), 6: 335 / 389, 4: 290 / 324, 3: 281 / 314,
    // 2: 328 / 361 (48 KB of LDS: one wavefront per SIMD); 100 k factors: 4 lanes 31.5, 3 lanes 32.5.
    // Round 6 (the result overlays the record: 928 B of LDS per factor, 8 wavefronts per CU at 3 lanes): 2 / 3 / 4 / 6 lanes
    // 252-254 / 249-255 / 256-258 / 315-317 (model 2: 276-282 / 269-272 / 282-283 / 334-337); 100 k factors 31.7 / 27.3 / 26.9-27.4 /
    // 31.0; 20 k 8.7 / 7.9-8.0 / 8.5 / 8.4: three at every size.
    int lpf = 3;
    (void)F;
#ifdef CPI_EXPERIMENTS
    if (expsw::packed_lpf()) lpf = expsw::packed_lpf();
#endif
    launch::factor_packed(model, lpf, a, packed, ctx->stream);
    CPI_HIP(ctx, hipGetLastError());
    return CPI_OK;
}

static int sqrt_information_impl(cpi_ctx *ctx, int64_t F, const double *P, double *R, bool packed) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (F < 0) return fail(ctx, CPI_ERR_INVALID, "cpi_sqrt_information_batch: negative size");
    if (F == 0) return CPI_OK;
    if (!P || !R) return fail(ctx, CPI_ERR_INVALID, "cpi_sqrt_information_batch: NULL argument");
    if (!grid_ok(F)) return fail(ctx, CPI_ERR_INVALID, "cpi_sqrt_information_batch: F exceeds 2^31 - 1 factors per call");
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    launch::sqrt_info((long long)F, P, R, packed, ctx->stream);
    CPI_HIP(ctx, hipGetLastError());
    return CPI_OK;
}
extern "C" int cpi_sqrt_information_batch(cpi_ctx *ctx, int64_t F, const double *P, double *sqrt_info) {
    return sqrt_information_impl(ctx, F, P, sqrt_info, false);
}
extern "C" int cpi_sqrt_information_packed_batch(cpi_ctx *ctx, int64_t F, const double *P_sym, double *R_tri) {
    return sqrt_information_impl(ctx, F, P_sym, R_tri, true);
}

static int factor_hessian_impl(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                               const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                               const double *states, int64_t S, const int32_t *idx_i, const int32_t *idx_j,
                               const double *sqrt_info, bool tri, double *hess) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (F == 0) return CPI_OK;
    if (!sqrt_info || !hess) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_hessian_batch: NULL argument");
    FactorArgs a;
    const int rc = factor_args(ctx, "cpi_factor_hessian_batch", model, grav, F, meas, lin, q_k_lin, states, S, idx_i, idx_j, a);
    if (rc != CPI_OK) return rc;
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    a.sqrt_info = sqrt_info; a.r_tri = tri ? 1 : 0;
    launch::factor_hessian(model, a, hess, ctx->stream);
    CPI_HIP(ctx, hipGetLastError());
    return CPI_OK;
}
extern "C" int cpi_factor_hessian_batch(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                                        const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                                        const double *states, int64_t S, const int32_t *idx_i, const int32_t *idx_j,
                                        const double *sqrt_info, double *hess) {
    return factor_hessian_impl(ctx, model, grav, F, meas, lin, q_k_lin, states, S, idx_i, idx_j, sqrt_info, false, hess);
}
extern "C" int cpi_factor_hessian_tri_batch(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                                            const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                                            const double *states, int64_t S, const int32_t *idx_i, const int32_t *idx_j,
                                            const double *R_tri, double *hess) {
    return factor_hessian_impl(ctx, model, grav, F, meas, lin, q_k_lin, states, S, idx_i, idx_j, R_tri, true, hess);
}

extern "C" int cpi_predict_batch(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                                 const cpi_outputs *meas, const double *states_i, int64_t S, const int32_t *idx_i,
                                 double *states_j) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (model != CPI_MODEL_V1 && model != CPI_MODEL_V2) return fail(ctx, CPI_ERR_INVALID, "cpi_predict_batch: model must be 1 or 2");
    if (F < 0) return fail(ctx, CPI_ERR_INVALID, "cpi_predict_batch: negative size");
    if (F == 0) return CPI_OK;
    if (!grav || !meas || !states_i || !states_j || !meas->DT || !meas->alpha || !meas->beta || !meas->q)
        return fail(ctx, CPI_ERR_INVALID, "cpi_predict_batch: NULL argument");
    if (!grid_ok(F)) return fail(ctx, CPI_ERR_INVALID, "cpi_predict_batch: F exceeds 2^31 - 1 factors per call");
    if (S <= 0 || (!idx_i && S < F)) return fail(ctx, CPI_ERR_INVALID, "cpi_predict_batch: S (number of states) must be >= 1, and >= F when idx_i is NULL");
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    PredictArgs a;
    memset(&a, 0, sizeof a);
    a.F = F;
    for (int i = 0; i < 3; i++) a.grav[i] = grav[i];
    a.meas = *meas; a.states_i = states_i; a.S = S; a.idx_i = idx_i; a.states_j = states_j;
    launch::predict(model, a, ctx->stream);
    CPI_HIP(ctx, hipGetLastError());
    return CPI_OK;
}

static const int OUT_N[kOutFields] = { 1, 3, 3, 4, 9, 9, 9, 9, 9, 9, 9, 225, CPI_TRI_DOUBLES };
static double **out_field(cpi_outputs *o, int k) {
    double **f[kOutFields] = { &o->DT, &o->alpha, &o->beta, &o->q, &o->J_q, &o->J_a, &o->J_b, &o->H_a, &o->H_b, &o->O_a, &o->O_b, &o->P, &o->P_sym };
    return f[k];
}
static double *out_field_c(const cpi_outputs *o, int k) { cpi_outputs t = *o; return *out_field(&t, k); }

extern "C" size_t cpi_outputs_slab_doubles(const cpi_outputs *mask, int64_t Wb) {
    if (!mask || Wb <= 0) return 0;
    size_t n = 0;
    for (int k = 0; k < kOutFields; k++) if (out_field_c(mask, k)) n += (size_t)OUT_N[k] * (size_t)Wb;
    return n;
}
extern "C" int cpi_outputs_bind_slab(const cpi_outputs *mask, int64_t Wb, double *slab, cpi_outputs *bound) {
    if (!mask || !bound || Wb < 0 || (!slab && Wb > 0)) return CPI_ERR_INVALID;
    cpi_outputs b;
    memset(&b, 0, sizeof b);
    size_t off = 0;
    for (int k = 0; k < kOutFields; k++)
        if (out_field_c(mask, k)) { *out_field(&b, k) = slab + off; off += (size_t)OUT_N[k] * (size_t)Wb; }
    *bound = b;
    return CPI_OK;
}

// ============================================================================================
// tiled layout: producers and the mean-only entry
// ============================================================================================
extern "C" int cpi_tile_windows(cpi_ctx *ctx, int64_t W, int32_t N, const double *knots, const int64_t *first,
                                const int32_t *count, double *tiles) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (W < 0 || N < 0) return fail(ctx, CPI_ERR_INVALID, "cpi_tile_windows: negative size");
    if (W == 0) return CPI_OK;
    if (!knots || !tiles) return fail(ctx, CPI_ERR_INVALID, "cpi_tile_windows: NULL argument");
    if (!grid_ok(W)) return fail(ctx, CPI_ERR_INVALID, "cpi_tile_windows: W exceeds 2^31 - 1 windows per call");
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    launch::tile_knots((long long)W, (int)N, knots, (const long long *)first, count, tiles, ctx->stream);
    CPI_HIP(ctx, hipGetLastError());
    return CPI_OK;
}
extern "C" int cpi_tile_knots(cpi_ctx *ctx, int64_t W, int32_t N, const double *knots, double *tiles) {
    return cpi_tile_windows(ctx, W, N, knots, nullptr, nullptr, tiles);
}
extern "C" int cpi_assemble_tiles(cpi_ctx *ctx, int64_t K, const double *stream, int64_t U, const double *update_times,
                                  int32_t N, double *tiles, int32_t *count) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (K < 0 || U < 0 || N < 0) return fail(ctx, CPI_ERR_INVALID, "cpi_assemble_tiles: negative size");
    if (U == 0) return CPI_OK;
    if (K == 0) return fail(ctx, CPI_ERR_INVALID, "cpi_assemble_tiles: the stream is empty");
    if (!stream || !update_times || !tiles || !count) return fail(ctx, CPI_ERR_INVALID, "cpi_assemble_tiles: NULL argument");
    if (!grid_ok(U)) return fail(ctx, CPI_ERR_INVALID, "cpi_assemble_tiles: U exceeds 2^31 - 1 windows per call");
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    AssembleArgs a;
    memset(&a, 0, sizeof a);
    a.K = K; a.stream = stream; a.U = U; a.update = update_times; a.N = N; a.tiles = tiles; a.count = count;
    a.ts = (long long)(N + 1) * 448; a.ss = 448;
    launch::assemble_tiles(a, ctx->stream);
    CPI_HIP(ctx, hipGetLastError());
    return CPI_OK;
}
extern "C" int cpi_preintegrate_tiled_batch(cpi_ctx *ctx, const cpi_params *prm, int64_t W, int32_t N, const double *tiles,
                                            const int32_t *count, const double *lin, const double *q_k_lin, const cpi_outputs *out) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (!prm || !out) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_tiled_batch: prm/out is NULL");
    if (prm->model != CPI_MODEL_V1 && prm->model != CPI_MODEL_V2) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_tiled_batch: model must be 1 or 2");
    if (W < 0 || N < 0) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_tiled_batch: negative size");
    if (W == 0) return CPI_OK;
    if (!tiles || !lin) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_tiled_batch: tiles/lin is NULL");
    if (prm->model == CPI_MODEL_V2 && !q_k_lin) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_tiled_batch: model 2 needs q_k_lin");
    if (!grid_ok(W)) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_tiled_batch: W exceeds 2^31 - 1 windows per call");
    if (out->J_q || out->J_a || out->J_b || out->H_a || out->H_b || out->O_a || out->O_b || out->P || out->P_sym)
        return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_tiled_batch: the tiled layout serves the mean outputs (DT, alpha, beta, q) only; "
                                          "Jacobians and covariance are FP64-bound, not HBM-bound: use cpi_preintegrate_batch");
    if (prm->lanes_per_window < 0 || prm->lanes_per_window > 8)
        return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_tiled_batch: lanes_per_window (here: wavefronts per tile) must be 0 (auto) or 1..8");
    if (!(out->DT || out->alpha || out->beta || out->q)) return CPI_OK;
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    TiledArgs a;
    memset(&a, 0, sizeof a);
    a.W = W; a.N = N; a.tiles = tiles; a.count = count; a.lin = lin; a.qk = q_k_lin; a.out = *out;
    for (int i = 0; i < 3; i++) a.grav[i] = prm->grav[i];
    a.ts = (long long)(N + 1) * 448; a.ss = 448;
    const long long nb = (W + 63) / 64;
#ifdef CPI_EXPERIMENTS
    a.dbg = expsw::blk_mode();
    if (a.dbg == 1) {   // CPI_AMD_PROBE_LDS = dynamic LDS bytes per wavefront, to pin the probe's occupancy (13312 -> 12 waves / CU)
        launch::tiled_fetch_probe(a, (size_t)expsw::probe_lds(), ctx->stream);
        CPI_HIP(ctx, hipGetLastError());
        return CPI_OK;
    }
#endif
    // wavefronts per tile.  Measured (MI355X, N = 50, us per launch, S = 1 / 2 / 3 / 4 / 8): 5 k windows 24.9 / 15.0 / 11.5 /
    // 10.3 / -, 10 k 25.2 / 15.4 / 11.9 / 10.8 / 12.4, 20 k 27.0 / 23.6 / 19.6 / 18.7 / 22.8, 30 k 28.6 / 25.1 / 22.2 / 21.3,
    // 50 k 33.0 / 34.4 / 34.8 / 34.9, 100 k 61.3 / 64.3 / 65.8 / 65.2: four (one per SIMD of the CU that owns the tile) while
    // the tiles do not fill the chip, one beyond.  prm->lanes_per_window (1..8) overrides the choice.
    int S = (nb < 640) ? std::max(1, std::min(4, (int)N / 4)) : 1;
    if (prm->lanes_per_window > 0) S = prm->lanes_per_window;
    CPI_HIP(ctx, launch::mean_tiled(prm->model, prm->imu_avg != 0, count != nullptr, S, a, ctx->stream, &ctx->big_lds_set));
    CPI_HIP(ctx, hipGetLastError());
    return CPI_OK;
}

// ============================================================================================
// device sets
// ============================================================================================
// Windows shard embarrassingly: rank r of n owns the contiguous block cpi_shard_bounds(W, r, n) and runs the ordinary
// entries on its own context; the ONE exchange step is the final gather of the output slabs to a root device.  RCCL is
// bound lazily (dlopen of librccl.so.1 at the first cpi_group_create with n > 1): single-GPU users never load it, and a
// process that already carries an RCCL (PyTorch) shares that copy.  One process drives all devices (ncclCommInitAll,
// rccl/rccl.h:236) -- the reference is a single process too; multi-process hosts (one rank per GPU, torch.distributed)
// use cpi_amd/dist.py, which issues the same send / recv pattern through ProcessGroupNCCL.
// The function-pointer types are decltype's of the prototypes in <rccl/rccl.h> and the datatype is its ncclFloat64, so a
// signature or enum drift is a compile error here, not a silent mismatch behind dlsym.
namespace {
struct Rccl {
    void *h = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
    std::mutex mu;
    bool load() {   // serialised: two host threads may create their first groups at the same time
        std::lock_guard<std::mutex> lock(mu);
        if (h) return true;
        // CPI_AMD_RCCL_LIB: an explicit library path (deployments with several ROCm installs; the test-suite points it at
        // tests/fake_rccl).  Read here, once, at the first n > 1 group -- never on a launch path.
        const char *forced = getenv("CPI_AMD_RCCL_LIB");
        std::string tried;
        if (forced && *forced) {
            h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            tried = forced;
        } else {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (h) break;
            }
            tried = "librccl.so.1";
        }
        if (!h) {
            const char *de = dlerror();   // dlerror() clears its state: call it ONCE
            err = "dlopen(" + tried + "): " + (de ? de : "not found");
            return false;
        }
#define CPI_SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(h, name)); \
        if (!field) { err = std::string("dlsym(") + name + "): symbol missing in " + tried; dlclose(h); h = nullptr; return false; }
        CPI_SYM(CommInitAll, "ncclCommInitAll") CPI_SYM(CommDestroy, "ncclCommDestroy") CPI_SYM(GroupStart, "ncclGroupStart")
        CPI_SYM(GroupEnd, "ncclGroupEnd") CPI_SYM(Send, "ncclSend") CPI_SYM(Recv, "ncclRecv") CPI_SYM(GetErrorString, "ncclGetErrorString")
#undef CPI_SYM
        return true;
    }
};
Rccl g_rccl;
constexpr int kMaxGroup = 16;   // ranks of one device set (a node has 8 GPUs)
}  // namespace

struct cpi_group {
    int n = 0;
    std::vector<cpi_ctx *> ctx;
    std::vector<hipStream_t> streams;   // owned
    std::vector<ncclComm_t> comms;      // empty when n == 1
    std::string err;
    // slab path of cpi_group_gather: the peers' slabs land here on the root's device before the unpack kernel places them
    double *staging = nullptr;
    size_t staging_cap = 0;             // doubles
    int staging_dev = -1;
    int last_gather_sends = 0;          // messages per peer of the last gather (1 = slab path); cpi_group_last_gather_messages
    // cpi_group_gather_chunk: a second non-blocking stream per device for the exchange (made at the first use), and one event per
    // device to order it behind / ahead of the compute stream
    std::vector<hipStream_t> xstreams;  // owned; empty until the first chunked gather
    std::vector<hipEvent_t> xev;
};
static thread_local std::string g_group_err;
static int gfail(cpi_group *g, int code, const std::string &msg) { if (g) g->err = msg; else g_group_err = msg; return code; }

extern "C" void cpi_shard_bounds(int64_t W, int rank, int n, int64_t *lo, int64_t *hi) {
    const int64_t per = n > 0 ? (W + n - 1) / n : W;
    const int64_t a = std::min<int64_t>(W, (int64_t)rank * per);
    if (lo) *lo = a;
    if (hi) *hi = std::min<int64_t>(W, a + per);
}
// sub-block `chunk` of `chunks` of rank's block: equal sub-block size on every rank (cper = ceil(ceil(W / n) / chunks))
extern "C" void cpi_shard_chunk_bounds(int64_t W, int rank, int n, int chunk, int chunks, int64_t *lo, int64_t *hi) {
    int64_t a, b;
    cpi_shard_bounds(W, rank, n, &a, &b);
    if (chunks > 1) {
        const int64_t per = n > 0 ? (W + n - 1) / n : W, cper = (per + chunks - 1) / chunks;
        const int64_t ca = std::min<int64_t>(b, a + (int64_t)chunk * cper);
        b = std::min<int64_t>(b, ca + cper);
        a = ca;
    }
    if (lo) *lo = a;
    if (hi) *hi = b;
}
extern "C" const char *cpi_group_last_error(const cpi_group *g) { return g ? g->err.c_str() : g_group_err.c_str(); }
extern "C" int cpi_group_size(const cpi_group *g) { return g ? g->n : 0; }
extern "C" cpi_ctx *cpi_group_ctx(cpi_group *g, int rank) { return (g && rank >= 0 && rank < g->n) ? g->ctx[rank] : nullptr; }
extern "C" int cpi_group_last_gather_messages(const cpi_group *g) { return g ? g->last_gather_sends : 0; }
extern "C" void cpi_group_destroy(cpi_group *g) {
    if (!g) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (int r = 0; r < g->n; r++) {
        if (r < (int)g->comms.size() && g->comms[r] && g_rccl.CommDestroy) g_rccl.CommDestroy(g->comms[r]);
        if (r < (int)g->ctx.size() && g->ctx[r]) {
            (void)hipSetDevice(g->ctx[r]->device);
            if (r < (int)g->xstreams.size() && g->xstreams[r]) (void)hipStreamDestroy(g->xstreams[r]);
            if (r < (int)g->xev.size() && g->xev[r]) (void)hipEventDestroy(g->xev[r]);
            if (r < (int)g->streams.size() && g->streams[r]) (void)hipStreamDestroy(g->streams[r]);
            cpi_ctx_destroy(g->ctx[r]);
        }
    }
    if (g->staging) { (void)hipSetDevice(g->staging_dev); (void)hipFree(g->staging); }
    if (prev >= 0) (void)hipSetDevice(prev);
    delete g;
}
static int group_create(int n, const int *devices, bool shared_device_for_tests, cpi_group **out) {
    if (!out) return gfail(nullptr, CPI_ERR_INVALID, "cpi_group_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return gfail(nullptr, CPI_ERR_NO_DEVICE, "cpi_group_create: no HIP device available (this library has no CPU fallback)");
    if (n <= 0 || n > kMaxGroup || (!shared_device_for_tests && n > ndev))
        return gfail(nullptr, CPI_ERR_INVALID, "cpi_group_create: n must be between 1 and the number of devices");
    std::vector<int> devs(n);
    for (int r = 0; r < n; r++) {
        devs[r] = devices ? devices[r] : r;
        if (devs[r] < 0 || devs[r] >= ndev) return gfail(nullptr, CPI_ERR_INVALID, "cpi_group_create: device index out of range");
        if (!shared_device_for_tests)
            for (int q = 0; q < r; q++) if (devs[q] == devs[r]) return gfail(nullptr, CPI_ERR_INVALID, "cpi_group_create: duplicate device");
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    cpi_group *g = new cpi_group();
    g->n = n;
    g->ctx.assign(n, nullptr); g->streams.assign(n, nullptr);
    for (int r = 0; r < n; r++) {
        hipError_t e = hipSetDevice(devs[r]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->streams[r], hipStreamNonBlocking);
        if (e != hipSuccess || cpi_ctx_create(devs[r], g->streams[r], &g->ctx[r]) != CPI_OK) {
            const std::string msg = std::string("cpi_group_create: device ") + std::to_string(devs[r]) + ": " + (e != hipSuccess ? hipGetErrorString(e) : cpi_last_error(nullptr));
            cpi_group_destroy(g);
            if (prev >= 0) (void)hipSetDevice(prev);
            return gfail(nullptr, CPI_ERR_HIP, msg);
        }
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    if (n > 1) {
        if (!g_rccl.load()) { const std::string m = "cpi_group_create: " + g_rccl.err; cpi_group_destroy(g); return gfail(nullptr, CPI_ERR_RCCL, m); }
        g->comms.assign(n, nullptr);
        const ncclResult_t rc = g_rccl.CommInitAll(g->comms.data(), n, devs.data());
        if (rc != ncclSuccess) { const std::string m = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(rc); cpi_group_destroy(g); return gfail(nullptr, CPI_ERR_RCCL, m); }
    }
    *out = g;
    return CPI_OK;
}
extern "C" int cpi_group_create(int n, const int *devices, cpi_group **out) { return group_create(n, devices, false, out); }
#ifdef CPI_TEST_HOOKS
// include/cpi_amd_test.h: n ranks that all live on ONE device -- the n > 1 code paths of the device set on a 1-GPU box.
// Real RCCL refuses duplicate devices in ncclCommInitAll; the test-suite binds tests/fake_rccl through CPI_AMD_RCCL_LIB.
// Not in the product library: the entry switches the duplicate-device guard of cpi_group_create off.
extern "C" int cpi_test_group_create_shared(int n, int device, cpi_group **out) {
    if (n <= 0 || n > kMaxGroup) return gfail(nullptr, CPI_ERR_INVALID, "cpi_test_group_create_shared: n out of range");
    std::vector<int> devs(n, device);
    return group_create(n, devs.data(), true, out);
}
#endif
extern "C" int cpi_group_synchronize(cpi_group *g) {
    if (!g) return gfail(nullptr, CPI_ERR_INVALID, "group is NULL");
    for (int r = 0; r < g->n; r++) {
        const int rc = cpi_ctx_synchronize(g->ctx[r]);
        if (rc != CPI_OK) return gfail(g, rc, cpi_last_error(g->ctx[r]));
    }
    if (!g->xstreams.empty()) {   // a chunked exchange that was not joined yet (cpi_group_gather_chunk)
        int prev = -1;
        (void)hipGetDevice(&prev);
        hipError_t e = hipSuccess;
        for (int r = 0; r < g->n && e == hipSuccess; r++) {
            e = hipSetDevice(g->ctx[r]->device);
            if (e == hipSuccess) e = hipStreamSynchronize(g->xstreams[r]);
        }
        if (prev >= 0) (void)hipSetDevice(prev);
        if (e != hipSuccess) return gfail(g, CPI_ERR_HIP, std::string("cpi_group_synchronize (exchange streams): ") + hipGetErrorString(e));
    }
    return CPI_OK;
}

// Is rank r's local output set ONE slab -- the wanted fields back to back, field-major over wb >= cnt windows
// (cpi_outputs_bind_slab)?  Returns its base, wb and the doubles to send (the last field only up to cnt windows).
static bool slab_of(const cpi_outputs &loc, const cpi_outputs &want, long long cnt, const double *&base, long long &wb, size_t &len) {
    int ks[kOutFields], nk = 0;
    for (int k = 0; k < kOutFields; k++) if (out_field_c(&want, k)) ks[nk++] = k;
    if (nk == 0) return false;
    const double *p0 = out_field_c(&loc, ks[0]);
    if (!p0) return false;
    wb = cnt;
    if (nk > 1) {
        const double *p1 = out_field_c(&loc, ks[1]);
        if (!p1 || p1 <= p0) return false;
        const long long d = (long long)(p1 - p0);
        if (d % OUT_N[ks[0]] != 0) return false;
        wb = d / OUT_N[ks[0]];
        if (wb < cnt) return false;
    }
    size_t off = 0;
    for (int i = 0; i < nk; i++) {
        if (out_field_c(&loc, ks[i]) != p0 + off) return false;
        if (i + 1 < nk) off += (size_t)OUT_N[ks[i]] * (size_t)wb;
    }
    base = p0;
    len = off + (size_t)OUT_N[ks[nk - 1]] * (size_t)cnt;
    return true;
}

// chunk < 0: the whole blocks, on the ranks' compute streams (cpi_group_gather).  chunk >= 0: sub-block `chunk` of `chunks` of every
// block, on the exchange streams, ordered behind the compute streams' position at this call; the last chunk joins
// (cpi_group_gather_chunk).
static int gather_impl(cpi_group *g, int root, int64_t W, int chunk, int chunks, const cpi_outputs *local, const cpi_outputs *root_out) {
    if (!g) return gfail(nullptr, CPI_ERR_INVALID, "group is NULL");
    if (root < 0 || root >= g->n || W < 0 || !local || !root_out) return gfail(g, CPI_ERR_INVALID, "cpi_group_gather: invalid argument");
    const bool chunked = chunk >= 0;
    if (chunked && (chunks < 1 || chunk >= chunks)) return gfail(g, CPI_ERR_INVALID, "cpi_group_gather_chunk: chunk must lie in [0, chunks)");
    int prev = -1;
    (void)hipGetDevice(&prev);
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore_{prev};
    const int n = g->n;
    if (chunked && g->xstreams.empty()) {
        g->xstreams.assign(n, nullptr); g->xev.assign(n, nullptr);
        for (int r = 0; r < n; r++)
            if (hipSetDevice(g->ctx[r]->device) != hipSuccess || hipStreamCreateWithFlags(&g->xstreams[r], hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&g->xev[r], hipEventDisableTiming) != hipSuccess)
                return gfail(g, CPI_ERR_HIP, "cpi_group_gather_chunk: creating the exchange streams failed");
    }
    auto stream_of = [&](int r) { return chunked ? g->xstreams[r] : g->ctx[r]->stream; };
    if (chunked) {   // the exchange of this sub-block starts where the compute streams stand NOW
        for (int r = 0; r < n; r++)
            if (hipSetDevice(g->ctx[r]->device) != hipSuccess || hipEventRecord(g->xev[r], g->ctx[r]->stream) != hipSuccess ||
                hipStreamWaitEvent(g->xstreams[r], g->xev[r], 0) != hipSuccess)
                return gfail(g, CPI_ERR_HIP, "cpi_group_gather_chunk: ordering the exchange stream behind the compute stream failed");
    }
    long long lo[kMaxGroup], cnt[kMaxGroup], wb[kMaxGroup];
    const double *base[kMaxGroup];
    size_t len[kMaxGroup], stride = 0;
    bool any_field = false;
    for (int k = 0; k < kOutFields; k++) any_field = any_field || out_field_c(root_out, k);
    if (!any_field) return CPI_OK;
    // every wanted field must exist in every non-empty block
    bool slabs = n > 1;
    for (int r = 0; r < n; r++) {
        int64_t a, b;
        if (chunked) cpi_shard_chunk_bounds(W, r, n, chunk, chunks, &a, &b); else cpi_shard_bounds(W, r, n, &a, &b);
        lo[r] = a; cnt[r] = b - a; wb[r] = 0; base[r] = nullptr; len[r] = 0;
        if (cnt[r] == 0) continue;
        for (int k = 0; k < kOutFields; k++)
            if (out_field_c(root_out, k) && !out_field_c(&local[r], k))
                return gfail(g, CPI_ERR_INVALID, "cpi_group_gather: a field wanted at the root is NULL in a rank's local outputs");
        if (r == root) continue;
        if (slabs && slab_of(local[r], *root_out, cnt[r], base[r], wb[r], len[r])) stride = std::max(stride, len[r]);
        else slabs = false;
    }
    hipStream_t rs = stream_of(root);
    // the slab path needs staging for n slabs on the root's device (grow-only; a re-allocation waits for the root's stream)
    if (slabs && stride > 0) {
        stride = (stride + 1) & ~(size_t)1;   // 16-byte aligned slabs
        if (g->staging_cap < stride * (size_t)n || g->staging_dev != g->ctx[root]->device) {
            if (g->staging) {
                if (hipSetDevice(g->staging_dev) != hipSuccess || hipDeviceSynchronize() != hipSuccess || hipFree(g->staging) != hipSuccess)
                    return gfail(g, CPI_ERR_HIP, "cpi_group_gather: releasing the staging buffer failed");
                g->staging = nullptr; g->staging_cap = 0;
            }
            if (hipSetDevice(g->ctx[root]->device) != hipSuccess || hipMalloc((void **)&g->staging, stride * (size_t)n * sizeof(double)) != hipSuccess)
                return gfail(g, CPI_ERR_HIP, "cpi_group_gather: allocating the root's staging buffer failed");
            g->staging_cap = stride * (size_t)n; g->staging_dev = g->ctx[root]->device;
        }
    }
    ncclResult_t rc = ncclSuccess;
    int hip_bad = 0, msgs = 0;
    if (n > 1) { rc = g_rccl.GroupStart(); if (rc != ncclSuccess) return gfail(g, CPI_ERR_RCCL, std::string("ncclGroupStart: ") + g_rccl.GetErrorString(rc)); }
    // the root's own block: device-to-device copies on its stream (unless it was computed in place)
    if (cnt[root] > 0) {
        for (int k = 0; k < kOutFields && !hip_bad; k++) {
            double *dst = out_field_c(root_out, k);
            if (!dst) continue;
            const double *src = out_field_c(&local[root], k);
            if (src != dst + (size_t)lo[root] * OUT_N[k]) {
                if (hipSetDevice(g->ctx[root]->device) != hipSuccess ||
                    hipMemcpyAsync(dst + (size_t)lo[root] * OUT_N[k], src, (size_t)cnt[root] * OUT_N[k] * sizeof(double), hipMemcpyDeviceToDevice, rs) != hipSuccess) hip_bad = 1;
            }
        }
    }
    // every peer sends straight to the root: one xGMI link per peer, no ring
    for (int r = 0; r < n && rc == ncclSuccess && !hip_bad; r++) {
        if (r == root || cnt[r] == 0) continue;
        if (slabs) {   // ONE message per peer: its whole slab into the root's staging area
            rc = g_rccl.Recv(g->staging + (size_t)r * stride, len[r], ncclFloat64, r, g->comms[root], rs);
            if (rc == ncclSuccess) rc = g_rccl.Send(base[r], len[r], ncclFloat64, root, g->comms[r], stream_of(r));
            msgs = 1;
        } else {       // separately allocated fields: one message per (peer, field), received in place
            int m = 0;
            for (int k = 0; k < kOutFields && rc == ncclSuccess; k++) {
                double *dst = out_field_c(root_out, k);
                if (!dst) continue;
                const size_t c = (size_t)cnt[r] * (size_t)OUT_N[k];
                rc = g_rccl.Recv(dst + (size_t)lo[r] * OUT_N[k], c, ncclFloat64, r, g->comms[root], rs);
                if (rc == ncclSuccess) rc = g_rccl.Send(out_field_c(&local[r], k), c, ncclFloat64, root, g->comms[r], stream_of(r));
                m++;
            }
            msgs = std::max(msgs, m);
        }
    }
    ncclResult_t rce = ncclSuccess;
    if (n > 1) rce = g_rccl.GroupEnd();
    if (hip_bad) return gfail(g, CPI_ERR_HIP, "cpi_group_gather: device-to-device copy of the root's own block failed");
    if (rc != ncclSuccess) return gfail(g, CPI_ERR_RCCL, std::string("ncclSend/ncclRecv: ") + g_rccl.GetErrorString(rc));
    if (rce != ncclSuccess) return gfail(g, CPI_ERR_RCCL, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(rce));
    if (slabs && msgs) {   // place the staged slabs: one launch on the root's stream, behind the receives
        long long c2[kMaxGroup];
        for (int r = 0; r < n; r++) c2[r] = (r == root) ? 0 : cnt[r];
        if (hipSetDevice(g->ctx[root]->device) != hipSuccess) return gfail(g, CPI_ERR_HIP, "cpi_group_gather: hipSetDevice(root) failed");
        launch::unpack_slabs(n, lo, c2, wb, g->staging, (long long)stride, *root_out, rs);
        if (hipGetLastError() != hipSuccess) return gfail(g, CPI_ERR_HIP, "cpi_group_gather: the unpack launch failed");
    }
    g->last_gather_sends = msgs;
    if (chunked && chunk == chunks - 1) {   // join: whatever is enqueued on the contexts from here on waits for the whole exchange
        for (int r = 0; r < n; r++)
            if (hipSetDevice(g->ctx[r]->device) != hipSuccess || hipEventRecord(g->xev[r], g->xstreams[r]) != hipSuccess ||
                hipStreamWaitEvent(g->ctx[r]->stream, g->xev[r], 0) != hipSuccess)
                return gfail(g, CPI_ERR_HIP, "cpi_group_gather_chunk: joining the exchange streams failed");
    }
    return CPI_OK;
}
extern "C" int cpi_group_gather(cpi_group *g, int root, int64_t W, const cpi_outputs *local, const cpi_outputs *root_out) {
    return gather_impl(g, root, W, -1, 1, local, root_out);
}
extern "C" int cpi_group_gather_chunk(cpi_group *g, int root, int64_t W, int chunk, int chunks, const cpi_outputs *local_chunk,
                                      const cpi_outputs *root_out) {
    return gather_impl(g, root, W, chunk, chunks, local_chunk, root_out);
}

// -------------------------------------------------------------------------------- test hook (include/cpi_amd_test.h)
#ifdef CPI_TEST_HOOKS
extern "C" int cpi_test_quat_ops(cpi_ctx *ctx, int32_t op, int64_t n, const double *in, double *out) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (op < 0 || op > 5 || n < 0) return fail(ctx, CPI_ERR_INVALID, "cpi_test_quat_ops: unknown op / negative size");
    if (n == 0) return CPI_OK;
    if (!in || !out) return fail(ctx, CPI_ERR_INVALID, "cpi_test_quat_ops: NULL argument");
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    launch::test_quat_ops((int)op, (long long)n, in, out, ctx->stream);
    CPI_HIP(ctx, hipGetLastError());
    return CPI_OK;
}
#endif

// ============================================================================================
// host-pointer variants
// ============================================================================================
namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};
}  // namespace
#define CPI_UP(buf, host, bytes)                                                          \
    do {                                                                                  \
        if (host) {                                                                       \
            CPI_HIP(ctx, hipMalloc(&buf.p, (bytes)));                                     \
            CPI_HIP(ctx, hipMemcpyAsync(buf.p, host, (bytes), hipMemcpyHostToDevice, ctx->stream)); \
        }                                                                                 \
    } while (0)

// Dense (and tiled) batches from host memory run as a three-stage pipeline over chunks of <= 65536 windows: upload of
// chunk i + 1 (copy stream), kernels of chunk i (the context's stream), download of chunk i - 1 (second copy stream) -- PCIe
// is full duplex, so with PINNED host buffers (cpi_host_alloc, hipHostMalloc, torch pin_memory) a call costs about
// max(upload, download, kernels) instead of their sum; with pageable memory the copies serialise in the runtime's own
// staging and the pipeline degenerates to the sum, minus the per-call hipMalloc / hipFree of the device staging, which
// the context now keeps (two slots, grow-only, released by cpi_ctx_destroy).  Measured (MI355X box, 1 M x 50, everything
// out = "V1 full", 2.9 GB up + 2.3 GB down): pinned 56 ms (92 GB/s both directions summed), pageable 112 ms; mean only
// 52 / 56 ms.  Page-locked bounce buffers + copy threads for pageable destinations were built and measured: no faster
// than the runtime's own path (112 ms) -- what costs is FRESH pageable output memory (first-touch page faults: 375-450 ms
// for the same call), so callers should re-use their output buffers.
struct HostPipe {
    hipStream_t up = nullptr, down = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    void *in[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};   // knots, count, lin, q_k_lin
    size_t in_cap[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    void *out[2][kOutFields] = {};
    size_t out_cap[2][kOutFields] = {};
};
static void host_pipe_destroy(HostPipe *hp) {
    if (!hp) return;
    for (int s = 0; s < 2; s++) {
        for (int k = 0; k < 4; k++) if (hp->in[s][k]) (void)hipFree(hp->in[s][k]);
        for (int k = 0; k < kOutFields; k++) if (hp->out[s][k]) (void)hipFree(hp->out[s][k]);
        if (hp->ev_in[s]) (void)hipEventDestroy(hp->ev_in[s]);
        if (hp->ev_done[s]) (void)hipEventDestroy(hp->ev_done[s]);
        if (hp->ev_out[s]) (void)hipEventDestroy(hp->ev_out[s]);
    }
    if (hp->up) (void)hipStreamDestroy(hp->up);
    if (hp->down) (void)hipStreamDestroy(hp->down);
    delete hp;
}
static int host_pipe_get(cpi_ctx *ctx) {
    if (ctx->pipe) return CPI_OK;
    HostPipe *hp = new HostPipe();
    ctx->pipe = hp;   // owned by the context from here on: a partial set-up is released by cpi_ctx_destroy
    CPI_HIP(ctx, hipStreamCreateWithFlags(&hp->up, hipStreamNonBlocking));
    CPI_HIP(ctx, hipStreamCreateWithFlags(&hp->down, hipStreamNonBlocking));
    for (int s = 0; s < 2; s++) {
        CPI_HIP(ctx, hipEventCreateWithFlags(&hp->ev_in[s], hipEventDisableTiming));
        CPI_HIP(ctx, hipEventCreateWithFlags(&hp->ev_done[s], hipEventDisableTiming));
        CPI_HIP(ctx, hipEventCreateWithFlags(&hp->ev_out[s], hipEventDisableTiming));
    }
    return CPI_OK;
}
static int host_pipe_reserve(cpi_ctx *ctx, void *&p, size_t &cap, size_t bytes) {
    if (bytes <= cap) return CPI_OK;
    if (p) { CPI_HIP(ctx, hipFree(p)); p = nullptr; cap = 0; }
    CPI_HIP(ctx, hipMalloc(&p, bytes));
    cap = bytes;
    return CPI_OK;
}
extern "C" void *cpi_host_alloc(size_t bytes) {
    void *p = nullptr;
    return (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess) ? p : nullptr;
}
extern "C" void cpi_host_free(void *p) { if (p) (void)hipHostFree(p); }

// tiled == false: knots[W][N+1][7];  tiled == true: tiles[ceil(W/64)][N+1][7][64] (chunks are whole tiles)
static int preintegrate_host_pipeline(cpi_ctx *ctx, const cpi_params *prm, int64_t W, int32_t N, const double *knots, bool tiled,
                                      const int32_t *count, const double *lin, const double *q_k_lin, const cpi_outputs *out) {
    int rc = host_pipe_get(ctx);
    if (rc != CPI_OK) return rc;
    HostPipe *hp = ctx->pipe;
    const int64_t nch = (W + 65535) / 65536;
    const int64_t Wc = std::min<int64_t>(W, (((W + nch - 1) / nch) + 63) / 64 * 64);   // balanced chunks, whole wavefronts / tiles
    const int nslots = nch > 1 ? 2 : 1;
    const size_t knot_bytes = (size_t)(N + 1) * 7 * sizeof(double);
    auto in_bytes = [&](int64_t wn) { return tiled ? (size_t)((wn + 63) / 64) * 64 * knot_bytes : (size_t)wn * knot_bytes; };
    cpi_outputs h = *out;
    for (int s = 0; s < nslots; s++) {
        const size_t need[4] = { in_bytes(Wc), count ? (size_t)Wc * sizeof(int32_t) : 0, (size_t)Wc * 6 * sizeof(double),
                                 q_k_lin ? (size_t)Wc * 4 * sizeof(double) : 0 };
        for (int k = 0; k < 4; k++)
            if ((rc = host_pipe_reserve(ctx, hp->in[s][k], hp->in_cap[s][k], need[k])) != CPI_OK) return rc;
        for (int k = 0; k < kOutFields; k++)
            if (*out_field(&h, k) && (rc = host_pipe_reserve(ctx, hp->out[s][k], hp->out_cap[s][k], (size_t)Wc * OUT_N[k] * sizeof(double))) != CPI_OK) return rc;
    }
    // after the first enqueue nothing may return before the three streams are idle: copies into the caller's memory are in flight
    std::string err;
    auto hip_ok = [&](hipError_t e, const char *what) { if (e != hipSuccess && err.empty()) err = std::string(what) + ": " + hipGetErrorString(e); return e == hipSuccess; };
    for (int64_t i = 0; i < nch && err.empty() && rc == CPI_OK; i++) {
        const int s = (int)(i & 1);
        const int64_t w0 = i * Wc, wn = std::min<int64_t>(Wc, W - w0);
        if (i >= 2 && !hip_ok(hipStreamWaitEvent(hp->up, hp->ev_done[s], 0), "hipStreamWaitEvent")) break;   // slot's inputs consumed
        if (!hip_ok(hipMemcpyAsync(hp->in[s][0], knots + (size_t)w0 * (N + 1) * 7, in_bytes(wn), hipMemcpyHostToDevice, hp->up), "upload knots")) break;
        if (count && !hip_ok(hipMemcpyAsync(hp->in[s][1], count + w0, (size_t)wn * sizeof(int32_t), hipMemcpyHostToDevice, hp->up), "upload count")) break;
        if (!hip_ok(hipMemcpyAsync(hp->in[s][2], lin + (size_t)w0 * 6, (size_t)wn * 6 * sizeof(double), hipMemcpyHostToDevice, hp->up), "upload lin")) break;
        if (q_k_lin && !hip_ok(hipMemcpyAsync(hp->in[s][3], q_k_lin + (size_t)w0 * 4, (size_t)wn * 4 * sizeof(double), hipMemcpyHostToDevice, hp->up), "upload q_k_lin")) break;
        if (!hip_ok(hipEventRecord(hp->ev_in[s], hp->up), "hipEventRecord")) break;
        if (!hip_ok(hipStreamWaitEvent(ctx->stream, hp->ev_in[s], 0), "hipStreamWaitEvent")) break;
        if (i >= 2 && !hip_ok(hipStreamWaitEvent(ctx->stream, hp->ev_out[s], 0), "hipStreamWaitEvent")) break;   // slot's outputs downloaded
        cpi_outputs d;
        memset(&d, 0, sizeof d);
        for (int k = 0; k < kOutFields; k++) if (*out_field(&h, k)) *out_field(&d, k) = (double *)hp->out[s][k];
        if (tiled)
            rc = cpi_preintegrate_tiled_batch(ctx, prm, wn, N, (const double *)hp->in[s][0], count ? (const int32_t *)hp->in[s][1] : nullptr,
                                              (const double *)hp->in[s][2], q_k_lin ? (const double *)hp->in[s][3] : nullptr, &d);
        else
            rc = cpi_preintegrate_batch(ctx, prm, wn, N, (const double *)hp->in[s][0], nullptr, count ? (const int32_t *)hp->in[s][1] : nullptr,
                                        (const double *)hp->in[s][2], q_k_lin ? (const double *)hp->in[s][3] : nullptr, &d);
        if (rc != CPI_OK) break;
        if (!hip_ok(hipEventRecord(hp->ev_done[s], ctx->stream), "hipEventRecord")) break;
        if (!hip_ok(hipStreamWaitEvent(hp->down, hp->ev_done[s], 0), "hipStreamWaitEvent")) break;
        for (int k = 0; k < kOutFields; k++)
            if (*out_field(&h, k) && !hip_ok(hipMemcpyAsync(*out_field(&h, k) + (size_t)w0 * OUT_N[k], hp->out[s][k], (size_t)wn * OUT_N[k] * sizeof(double),
                                                            hipMemcpyDeviceToHost, hp->down), "download")) break;
        if (!err.empty()) break;
        if (!hip_ok(hipEventRecord(hp->ev_out[s], hp->down), "hipEventRecord")) break;
    }
    const hipError_t e1 = hipStreamSynchronize(hp->up), e2 = hipStreamSynchronize(ctx->stream), e3 = hipStreamSynchronize(hp->down);
    if (rc != CPI_OK) return rc;   // message already set by the device-pointer entry
    if (!err.empty()) return fail(ctx, CPI_ERR_HIP, "cpi_preintegrate_batch_host: " + err);
    hip_ok(e1, "hipStreamSynchronize(upload)"); hip_ok(e2, "hipStreamSynchronize"); hip_ok(e3, "hipStreamSynchronize(download)");
    if (!err.empty()) return fail(ctx, CPI_ERR_HIP, "cpi_preintegrate_batch_host: " + err);
    return CPI_OK;
}

extern "C" int cpi_preintegrate_tiled_batch_host(cpi_ctx *ctx, const cpi_params *prm, int64_t W, int32_t N, const double *tiles,
                                                 const int32_t *count, const double *lin, const double *q_k_lin, const cpi_outputs *out) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (!prm || !out || !tiles || !lin) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_tiled_batch_host: NULL argument");
    if (W <= 0) return W == 0 ? CPI_OK : fail(ctx, CPI_ERR_INVALID, "negative size");
    if (N < 0) return fail(ctx, CPI_ERR_INVALID, "negative size");
    if (out->J_q || out->J_a || out->J_b || out->H_a || out->H_b || out->O_a || out->O_b || out->P || out->P_sym)
        return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_tiled_batch_host: the tiled layout serves the mean outputs (DT, alpha, beta, q) only");
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    return preintegrate_host_pipeline(ctx, prm, W, N, tiles, true, count, lin, q_k_lin, out);
}

extern "C" int cpi_preintegrate_batch_host(cpi_ctx *ctx, const cpi_params *prm, int64_t W, int32_t N,
                                           const double *knots, const int64_t *first, const int32_t *count,
                                           int64_t n_knots, const double *lin, const double *q_k_lin,
                                           const cpi_outputs *out) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (!prm || !out || !knots || !lin) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_batch_host: NULL argument");
    if (W <= 0) return W == 0 ? CPI_OK : fail(ctx, CPI_ERR_INVALID, "negative size");
    if (N < 0) return fail(ctx, CPI_ERR_INVALID, "negative size");
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    if (!first) return preintegrate_host_pipeline(ctx, prm, W, N, knots, false, count, lin, q_k_lin, out);
    // ragged windows share one knot stream: staged whole (one-off calls; the stream is usually small)
    DevBuf dk, df, dc, dl, dq, dout[kOutFields];
    CPI_UP(dk, knots, (size_t)n_knots * 7 * sizeof(double));
    CPI_UP(df, first, (size_t)W * sizeof(int64_t));
    CPI_UP(dc, count, (size_t)W * sizeof(int32_t));
    CPI_UP(dl, lin, (size_t)W * 6 * sizeof(double));
    CPI_UP(dq, q_k_lin, (size_t)W * 4 * sizeof(double));
    cpi_outputs d = *out, h = *out;
    for (int k = 0; k < kOutFields; k++)
        if (*out_field(&h, k)) {
            CPI_HIP(ctx, hipMalloc(&dout[k].p, (size_t)W * OUT_N[k] * sizeof(double)));
            *out_field(&d, k) = (double *)dout[k].p;
        }
    int rc = cpi_preintegrate_batch(ctx, prm, W, N, (const double *)dk.p, (const int64_t *)df.p, (const int32_t *)dc.p,
                                    (const double *)dl.p, (const double *)dq.p, &d);
    if (rc != CPI_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    for (int k = 0; k < kOutFields; k++)
        if (*out_field(&h, k))
            CPI_HIP(ctx, hipMemcpyAsync(*out_field(&h, k), dout[k].p, (size_t)W * OUT_N[k] * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    CPI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CPI_OK;
}

// The stream entry from HOST memory: what a GraphSolver-shaped caller holds (its IMU deque as one array, the update times of the
// states it creates, one linearisation point per window) -> the measurements of every window, in host memory.  One-off
// staging (the stream is uploaded once, whole); the windows are cut on the device.
extern "C" int cpi_preintegrate_stream_host(cpi_ctx *ctx, const cpi_params *prm, int64_t K, const double *stream, int64_t U,
                                            const double *update_times, int32_t N, const double *lin, const double *q_k_lin,
                                            const cpi_outputs *out, int32_t *count) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (K < 0 || U < 0 || N < 0) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_stream_host: negative size");
    if (U == 0) return CPI_OK;
    if (K == 0) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_stream_host: the stream is empty");
    if (!prm || !out || !stream || !update_times || !lin) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_stream_host: NULL argument");
    if (prm->model == CPI_MODEL_V2 && !q_k_lin) return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_stream_host: model 2 needs q_k_lin");
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    DevBuf ds, du, dl, dq, dw, dout[kOutFields];
    CPI_UP(ds, stream, (size_t)K * 7 * sizeof(double));
    CPI_UP(du, update_times, (size_t)U * sizeof(double));
    CPI_UP(dl, lin, (size_t)U * 6 * sizeof(double));
    CPI_UP(dq, q_k_lin, (size_t)U * 4 * sizeof(double));
    CPI_HIP(ctx, hipMalloc(&dw.p, cpi_stream_workspace_bytes(U)));
    cpi_outputs d = *out, h = *out;
    for (int k = 0; k < kOutFields; k++)
        if (*out_field(&h, k)) {
            CPI_HIP(ctx, hipMalloc(&dout[k].p, (size_t)U * OUT_N[k] * sizeof(double)));
            *out_field(&d, k) = (double *)dout[k].p;
        }
    const int rc = cpi_preintegrate_stream(ctx, prm, K, (const double *)ds.p, U, (const double *)du.p, N, (const double *)dl.p,
                                           (const double *)dq.p, dw.p, &d);
    if (rc != CPI_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    for (int k = 0; k < kOutFields; k++)
        if (*out_field(&h, k))
            CPI_HIP(ctx, hipMemcpyAsync(*out_field(&h, k), dout[k].p, (size_t)U * OUT_N[k] * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (count) CPI_HIP(ctx, hipMemcpyAsync(count, cpi_stream_counts(dw.p, U), (size_t)U * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    CPI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CPI_OK;
}

extern "C" int cpi_factor_eval_batch_host(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                                          const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                                          const double *states, int64_t S, const int32_t *idx_i,
                                          const int32_t *idx_j, double *err, double *H1, double *H2) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (!meas || !lin || !states || !err || !grav) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_batch_host: NULL argument");
    if (F <= 0) return F == 0 ? CPI_OK : fail(ctx, CPI_ERR_INVALID, "negative size");
    // host pointers: the indices can be (and are) validated here; the device-pointer entries clamp them instead
    if (S <= 0 || (!idx_i && S < F) || (!idx_j && S < F + 1)) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_batch_host: too few states");
    for (int64_t f = 0; f < F; f++)
        if ((idx_i && (idx_i[f] < 0 || idx_i[f] >= S)) || (idx_j && (idx_j[f] < 0 || idx_j[f] >= S)))
            return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_batch_host: state index out of range at factor " + std::to_string(f));
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    DevBuf dl, dq, ds, di, dj, de, dh1, dh2, dm[11];
    cpi_outputs hm = *meas, d;
    memset(&d, 0, sizeof d);
    for (int k = 0; k < 11; k++)
        if (*out_field(&hm, k)) {
            CPI_UP(dm[k], *out_field(&hm, k), (size_t)F * OUT_N[k] * sizeof(double));
            *out_field(&d, k) = (double *)dm[k].p;
        }
    CPI_UP(dl, lin, (size_t)F * 6 * sizeof(double));
    CPI_UP(dq, q_k_lin, (size_t)F * 4 * sizeof(double));
    CPI_UP(ds, states, (size_t)S * 16 * sizeof(double));
    CPI_UP(di, idx_i, (size_t)F * sizeof(int32_t));
    CPI_UP(dj, idx_j, (size_t)F * sizeof(int32_t));
    CPI_HIP(ctx, hipMalloc(&de.p, (size_t)F * 15 * sizeof(double)));
    if (H1) CPI_HIP(ctx, hipMalloc(&dh1.p, (size_t)F * 225 * sizeof(double)));
    if (H2) CPI_HIP(ctx, hipMalloc(&dh2.p, (size_t)F * 225 * sizeof(double)));
    int rc = cpi_factor_eval_batch(ctx, model, grav, F, &d, (const double *)dl.p, (const double *)dq.p, (const double *)ds.p, S,
                                   (const int32_t *)di.p, (const int32_t *)dj.p, (double *)de.p, (double *)dh1.p, (double *)dh2.p);
    if (rc != CPI_OK) return rc;
    CPI_HIP(ctx, hipMemcpyAsync(err, de.p, (size_t)F * 15 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (H1) CPI_HIP(ctx, hipMemcpyAsync(H1, dh1.p, (size_t)F * 225 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (H2) CPI_HIP(ctx, hipMemcpyAsync(H2, dh2.p, (size_t)F * 225 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    CPI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CPI_OK;
}
