// cpi_args.hpp -- kernel argument blocks and the launcher interface between the translation units of libcpi_amd.so.
//
// The library is four translation units, compiled in parallel by cpi_amd/build.py and linked into one shared object:
//   cpi_mean.hip    cpi_mean_kernel / cpi_mean_tiled_kernel / cpi_tile_*_kernel       (cpi_mean_kernels.hpp)
//   cpi_cov.hip     cpi_cov_kernel<1|2> / cpi_forster_kernel                           (cpi_cov_kernels.hpp)
//   cpi_factor.hip  evaluateError sweeps, square-root information, Hessian blocks, state prediction
//                                                                                      (cpi_factor_kernels.hpp)
//   cpi_abi.hip     the C-ABI of include/cpi_amd.h: argument checks, launch heuristics, device sets (RCCL), the
//                   host-pointer pipeline.  No kernels.
// A kernel TU exports plain host functions (namespace cpi::launch) that pick the template instantiation and enqueue it
// on the given stream; nothing else crosses a TU boundary (no relocatable device code).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "../../include/cpi_amd.h"

namespace cpi {

struct PreArgs {
    long long W;
    int N;
    const double *knots;
    const long long *first;
    const int *count;
    const double *lin;
    const double *qk;
    // Windows cut out of ONE IMU stream in flight (cpi_preintegrate_stream; GraphSolver_IMU.cpp:50-69): knots = the stream,
    // first[w] = the front reading of window w, count[w] = whole intervals + the partial tail interval; knot 0 of the window
    // carries the stamp tstart[w] instead of its own, and when tend[w] is not NaN the last interval is the tail
    // [stamp of the last real knot, tend[w]] with that knot's reading held.  Both NULL: plain knots.
    const double *tstart;
    const double *tend;
    // FUSED cut (mean-only requests of cpi_preintegrate_stream; cpi_mean_kernel<..., CUT = 2>): knots = the stream of K
    // readings, update[W] = the update times; every wavefront finds where the reference's deque stands for its own windows
    // (the arithmetic of cpi_cut_windows_kernel, in registers) and writes the TRUE interval count to count_out[W] --
    // first / count / tstart / tend stay NULL and no cut kernel runs.  K > 0 also tells a CUT = 1 launch where the stream
    // ends (fast addressing: a wavefront whose furthest read stays inside the stream needs no per-element pointers).
    const double *update;
    long long K;
    int *count_out;
    double grav[3];
    double q4[4];      // sigma^2 of the four diagonal blocks of Q_c (CpiBase.h:54-57)
    int write_means;   // kernel writes DT/alpha/beta/q
    int write_jac;     // kernel writes the Jacobians it owns
    int dbg;           // development switches of the experimental kernels (0 in every shipped path)
    cpi_outputs out;
};

// tiles[b][s][k][i] = field k (t, w, a) of knot s of window 64 b + i (include/cpi_amd.h: cpi_preintegrate_tiled_batch)
struct TiledArgs {
    long long W;
    int N;
    const double *tiles;
    const int *count;
    const double *lin;
    const double *qk;
    double grav[3];
    cpi_outputs out;
    int dbg;            // CPI_EXPERIMENTS builds only: 1 = fetch without arithmetic
    long long ts, ss;   // doubles between consecutive tiles / consecutive steps of a tile
};

struct FactorArgs {
    long long F;
    double grav[3];
    cpi_outputs meas;
    const double *lin;
    const double *qk;
    const double *states;
    long long S;               // number of states: indices are clamped into [0, S) (no out-of-bounds read whatever idx holds)
    const int *idx_i;
    const int *idx_j;
    double *err;
    double *H1;
    double *H2;
    const double *sqrt_info;   // optional [F][225] upper-triangular R: outputs are whitened (R err, R H1, R H2)
    int r_tri;                 // sqrt_info is the packed upper triangle [F][120] (include/cpi_amd.h: CPI_TRI_INDEX)
};

struct PredictArgs {
    long long F;
    double grav[3];
    cpi_outputs meas;
    const double *states_i;
    long long S;
    const int *idx_i;
    double *states_j;
};

// Window assembly on the device (cpi_assemble_tiles, include/cpi_amd.h): one IMU stream cut at update times straight into
// the tiled layout.
struct AssembleArgs {
    long long K;            // knots of the stream
    const double *stream;   // [K][7]
    long long U;            // windows
    const double *update;   // [U] update times, non-decreasing
    int N;                  // rows of a tile - 1 (>= the largest count)
    double *tiles;          // [ceil(U/64)][N+1][7][64]
    int *count;             // [U]
    long long ts, ss;
};

namespace launch {
// ---- cpi_mean.hip
bool mean_lanes_supported(int L);
int mean_lane_choices(const int **list);   // the supported L values, ascending
void mean(int model, bool jac, bool avg, int L, const PreArgs &a, hipStream_t st);
// S wavefronts per tile (1 = one wavefront owns the tile; > 1 = SPLIT); big_lds_set: per-context bit set of the
// instantiations whose dynamic-LDS limit was already raised
hipError_t mean_tiled(int model, bool avg, bool counted, int S, const TiledArgs &a, hipStream_t st, unsigned *big_lds_set);
void tile_knots(long long W, int N, const double *knots, const long long *first, const int *count, double *tiles, hipStream_t st);
void assemble_tiles(const AssembleArgs &a, hipStream_t st);
void cut_windows(long long K, const double *stream, long long U, const double *update, int N, long long *first, int *count,
                 double *tstart, double *tend, hipStream_t st);
// ---- cpi_cov.hip
void cov(int model, bool avg, const PreArgs &a, hipStream_t st);
void forster(const PreArgs &a, hipStream_t st);
// ---- cpi_factor.hip
void factor(int model, bool whiten, int lpf, const FactorArgs &a, hipStream_t st);            // lpf 16 | 8 | 4
void factor_packed(int model, int lpf, const FactorArgs &a, double *packed, hipStream_t st);  // lpf 2 | 3 | 4 | 6 | 8
void factor_hessian(int model, const FactorArgs &a, double *hess, hipStream_t st);
void sqrt_info(long long F, const double *P, double *R, bool packed, hipStream_t st);   // packed: P_sym [F][120] -> R_tri [F][120]
void predict(int model, const PredictArgs &a, hipStream_t st);
#ifdef CPI_TEST_HOOKS
void test_quat_ops(int op, long long n, const double *in, double *out, hipStream_t st);   // libcpi_amd_test.so only
#endif
void unpack_slabs(int n, const long long *lo, const long long *cnt, const long long *wb, const double *staging, long long stride,
                  const cpi_outputs &root_out, hipStream_t st);
#ifdef CPI_EXPERIMENTS
// measurement-only kernels (cpi_mean_experimental.hpp; tools/exp/): never part of the default build
struct MeanDmaCfg { int kc, s, aligned; };
long long mean_dma(int model, const MeanDmaCfg &c, bool avg, const PreArgs &a, hipStream_t st);   // leading windows handled
bool mean_blk(int model, int L, bool avg, const PreArgs &a, hipStream_t st);
long long mean_line(int model, bool avg, const PreArgs &a, hipStream_t st);                       // leading windows handled
void tiled_fetch_probe(const TiledArgs &a, size_t lds, hipStream_t st);
#endif
}  // namespace launch
}  // namespace cpi
