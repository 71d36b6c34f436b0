// cpi_mean.hip -- translation unit of the mean (+ analytic Jacobian) kernels: cpi_mean_kernel (dense / ragged layouts),
// cpi_mean_tiled_kernel (tiled layout), the tile producers, and their launchers (cpi_args.hpp: cpi::launch).
// Replaces CpiV1.h:67-259 / CpiV2.h:88-305; window assembly GraphSolver_IMU.cpp:50-69.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "cpi_args.hpp"
#include "cpi_math.hpp"

using namespace cpi;

#include "cpi_device_util.hpp"
#include "cpi_mean_kernels.hpp"
#ifdef CPI_EXPERIMENTS
#include "cpi_mean_experimental.hpp"
#endif

namespace cpi {
namespace launch {

static const int kMeanLanes[] = {1, 2, 3, 4, 5, 6, 8, 12, 16, 32, 64};
int mean_lane_choices(const int **list) { *list = kMeanLanes; return (int)(sizeof kMeanLanes / sizeof kMeanLanes[0]); }
bool mean_lanes_supported(int L) {
    for (int c : kMeanLanes) if (c == L) return true;
    return false;
}

// One lane per window, mean-only: from this many windows the three-knots-per-chunk instantiation (cpi_mean_kernel<..., BIG>:
// 32-bit staging offsets from a wave-uniform base, flat LDS tile, two wavefronts per SIMD at ~200 registers) replaces the
// two-knot one.  Admission rule of the 32-bit offsets: every lane-segment of a wavefront within 2^32 bytes above the wavefront's
// lowest -- any 64 stream windows of a stream of < 2^26 readings, any 64 consecutive windows of the dense layout (N <= 65 535);
// a CSR `first` array (arbitrary addresses) and per-window counts keep the two-knot kernel.
// Measured on one box, alternating (profiles/r04_mean_chunk_ab.md, part D): the stream entry gains from 65 k windows (100 k:
// 74-76 -> 72-73 us, 1 M x 51: 668-671 -> 655-658; jittered update times, every wavefront on the per-element path: 772 -> 718);
// the dense layout at 1 M x 50 644-648 -> 629-630 us (1 M x 100: 1 230 -> 1 210-1 219), model 2 671-673 -> 661-663, both EQUAL
// within the noise from 65 k to 500 k windows -- hence the three thresholds.  What it does NOT buy at two wavefronts per SIMD is
// HBM traffic: FETCH_SIZE of the 1 M launches is 1.37 x (dense) / 1.48 x (stream) algorithmic at 8 wavefronts per CU, 1.14 /
// 1.17 x at 6, 1.10 / 1.12 x at 4-5 -- the lines a chunk leaves half-read survive in the L2 only while few wavefronts stream
// through it -- and the time goes the other way (dense 628 / 656 / 694 us, stream 673 / 685 / 714 us at 8 / 6 / 4 per CU).
// CPI_MEAN_BIG_LDS_PAD (bytes of unused dynamic LDS per wavefront: 16 000 = 6 per CU, 29 000 = 4) is that knob; shipped: 0,
// the fastest point.
#ifndef CPI_MEAN_BIG_W
#define CPI_MEAN_BIG_W 100000
#endif
#ifndef CPI_MEAN_BIG_W_DENSE
#define CPI_MEAN_BIG_W_DENSE 700000
#endif
#ifndef CPI_MEAN_BIG_W_M2
#define CPI_MEAN_BIG_W_M2 700000
#endif
#ifndef CPI_MEAN_BIG_LDS_PAD
#define CPI_MEAN_BIG_LDS_PAD 0
#endif
// Round 6, the reference's own window lengths (imurate / camrate = 10 and 20, synthetic_test.launch:27-28): at N = 10 three knots per
// chunk is 3 1/3 chunks with a padded step, and the two-knot kernel (5 chunks exactly) is 4-5 % faster on the dense layout (1 M x 10:
// 158-159 vs 164-167 us; model 2 164 vs 170) and on stream windows (197 vs 206-208); at N = 20 the two are equal within the run-to-run
// noise (268-285 vs 280; stream 315-317 vs 308-311).  Five knots per chunk (256 registers + 32 B of scratch) wins only the N = 10
// stream (191 us) and is not instantiated.  Hence: BIG from 16 intervals per window (profiles/r06_short_windows.md).
#ifndef CPI_MEAN_BIG_NMIN
#define CPI_MEAN_BIG_NMIN 16
#endif
template <int MODEL, bool JAC, bool AVG>
static void launch_mean_L(int L, const PreArgs &a, hipStream_t st) {
    // 0: plain knots; 1: windows cut by cpi_cut_windows_kernel (workspace route); 2: the wavefront cuts its own windows
    // (mean-only requests of cpi_preintegrate_stream; no analytic-Jacobian instantiations -- the caller never asks)
    const int cut = a.update != nullptr ? 2 : (a.tstart != nullptr ? 1 : 0);
    if constexpr (!JAC) {
        const long long wmin = MODEL == 2 ? (long long)CPI_MEAN_BIG_W_M2 : (cut != 0 ? (long long)CPI_MEAN_BIG_W : (long long)CPI_MEAN_BIG_W_DENSE);
        const bool admitted = cut != 0 ? (a.K > 0 && a.K < (1ll << 26)) : (a.first == nullptr && a.count == nullptr);
        if (L == 1 && admitted && a.W >= wmin && a.N >= CPI_MEAN_BIG_NMIN) {
            const unsigned nb = (unsigned)((a.W + 63) / 64);
            const size_t pad = CPI_MEAN_BIG_LDS_PAD;   // unused dynamic LDS: caps the wavefronts per CU (see above)
            if (cut == 2) hipLaunchKernelGGL((cpi_mean_kernel<MODEL, false, AVG, 1, 2, true>), dim3(nb), dim3(64), pad, st, a);
            else if (cut == 1) hipLaunchKernelGGL((cpi_mean_kernel<MODEL, false, AVG, 1, 1, true>), dim3(nb), dim3(64), pad, st, a);
            else hipLaunchKernelGGL((cpi_mean_kernel<MODEL, false, AVG, 1, 0, true>), dim3(nb), dim3(64), pad, st, a);
            return;
        }
    }
#define CPI_LAUNCH_L(LL)                                                                         \
    case LL: {                                                                                   \
        const long long nb = (a.W + (64 / LL) - 1) / (64 / LL);                                  \
        if (cut == 2) {                                                                          \
            if constexpr (!JAC) hipLaunchKernelGGL((cpi_mean_kernel<MODEL, JAC, AVG, LL, 2>), dim3((unsigned)nb), dim3(64), 0, st, a); \
        } else if (cut == 1) hipLaunchKernelGGL((cpi_mean_kernel<MODEL, JAC, AVG, LL, 1>), dim3((unsigned)nb), dim3(64), 0, st, a); \
        else hipLaunchKernelGGL((cpi_mean_kernel<MODEL, JAC, AVG, LL, 0>), dim3((unsigned)nb), dim3(64), 0, st, a); \
    } break;
    if constexpr (MODEL == 2 && JAC) { switch (L) { CPI_LAUNCH_L(1) default: break; } } else
    switch (L) {
        CPI_LAUNCH_L(1) CPI_LAUNCH_L(2) CPI_LAUNCH_L(3) CPI_LAUNCH_L(4) CPI_LAUNCH_L(5) CPI_LAUNCH_L(6) CPI_LAUNCH_L(8)
        CPI_LAUNCH_L(12) CPI_LAUNCH_L(16) CPI_LAUNCH_L(32) CPI_LAUNCH_L(64)
        default: break;
    }
#undef CPI_LAUNCH_L
}
template <int MODEL>
static void launch_mean_M(bool jac, bool avg, int L, const PreArgs &a, hipStream_t st) {
    if (jac) { if (avg) launch_mean_L<MODEL, true, true>(L, a, st); else launch_mean_L<MODEL, true, false>(L, a, st); }
    else     { if (avg) launch_mean_L<MODEL, false, true>(L, a, st); else launch_mean_L<MODEL, false, false>(L, a, st); }
}
void mean(int model, bool jac, bool avg, int L, const PreArgs &a, hipStream_t st) {
    if (model == CPI_MODEL_V2) launch_mean_M<2>(jac, avg, L, a, st); else launch_mean_M<1>(jac, avg, L, a, st);
}

hipError_t mean_tiled(int model, bool avg, bool counted, int S, const TiledArgs &a, hipStream_t st, unsigned *big_lds_set) {
    const unsigned nb = (unsigned)((a.W + 63) / 64);
    const size_t lds = (size_t)(S - 1) * (model == CPI_MODEL_V2 ? 34 : 16) * 64 * sizeof(double);
    // more than 64 KB of dynamic LDS (model 2, S >= 5) needs the kernel's limit raised once per device
#define CPI_TILED2(M, AV, C)                                                                                      \
    do {                                                                                                          \
        if (S > 1) {                                                                                              \
            const unsigned bit = 1u << ((M - 1) * 4 + (AV ? 2 : 0) + (C ? 1 : 0));                                \
            if (lds > 65536 && !(*big_lds_set & bit)) {                                                           \
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&cpi_mean_tiled_kernel<M, AV, C, true>), \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 7 * 34 * 64 * 8);  \
                if (e != hipSuccess) return e;                                                                    \
                *big_lds_set |= bit;                                                                              \
            }                                                                                                     \
            hipLaunchKernelGGL((cpi_mean_tiled_kernel<M, AV, C, true>), dim3(nb), dim3(64 * S), lds, st, a);      \
        } else hipLaunchKernelGGL((cpi_mean_tiled_kernel<M, AV, C, false>), dim3(nb), dim3(64), 0, st, a);        \
    } while (0)
    if (model == CPI_MODEL_V1) {
        if (counted) { if (avg) CPI_TILED2(1, true, true); else CPI_TILED2(1, false, true); }
        else         { if (avg) CPI_TILED2(1, true, false); else CPI_TILED2(1, false, false); }
    } else {
        if (counted) { if (avg) CPI_TILED2(2, true, true); else CPI_TILED2(2, false, true); }
        else         { if (avg) CPI_TILED2(2, true, false); else CPI_TILED2(2, false, false); }
    }
#undef CPI_TILED2
    return hipSuccess;
}

void tile_knots(long long W, int N, const double *knots, const long long *first, const int *count, double *tiles, hipStream_t st) {
    const long long total = ((W + 63) / 64) * (long long)(N + 1) * 448;
    const unsigned nb = (unsigned)std::min<long long>((total + 255) / 256, 256 * 64);
    // tile-major.  (Step-major -- tiles[N+1][ceil(W/64)][7][64], all resident wavefronts reading one moving window -- was
    // measured: 557 vs 571 us per 1 M x 50, 58.2 vs 61.1 us per 100 k, stream alone 472 vs 481 us: not worth a second contract.)
    const long long ts = (long long)(N + 1) * 448, ss = 448;
    hipLaunchKernelGGL(cpi_tile_knots_kernel, dim3(nb), dim3(256), 0, st, W, N, knots, first, count, tiles, ts, ss);
}

void cut_windows(long long K, const double *stream, long long U, const double *update, int N, long long *first, int *count,
                 double *tstart, double *tend, hipStream_t st) {
    (void)N;
    hipLaunchKernelGGL(cpi_cut_windows_kernel, dim3((unsigned)((U + 255) / 256)), dim3(256), 0, st, K, stream, U, update, first, count, tstart, tend);
}

void assemble_tiles(const AssembleArgs &a, hipStream_t st) {
    const unsigned nb = (unsigned)((a.U + 63) / 64);
    hipLaunchKernelGGL(cpi_assemble_tiles_kernel, dim3(nb), dim3(64), 0, st, a);
}

#ifdef CPI_EXPERIMENTS
template <int KC, bool ALIGNED>
static long long mean_dma_safe_blocks(long long W, int N) {
    typedef DmaGeom<KC, ALIGNED> G;
    const long long wstride = (long long)(N + 1) * 56;
    const long long nst = (N + KC - 1) / KC;
    long long nb = W / 64;
    // furthest byte a block touches: window (b*64 + NI*WPI - 1), knot 1 + nst*KC, plus 16 bytes of alignment slack
    while (nb > 0 && ((nb - 1) * 64 + (long long)G::NI * G::WPI - 1) * wstride + 56 + nst * KC * 56 + 16 > W * wstride) --nb;
    return nb;
}
template <int MODEL, int KC, int S, bool ALIGNED>
static long long launch_mean_dma_one(bool avg, const PreArgs &a, hipStream_t st) {
    const long long nb = mean_dma_safe_blocks<KC, ALIGNED>(a.W, a.N);
    if (nb <= 0) return 0;
    if (avg) hipLaunchKernelGGL((cpi_mean_dma_kernel<MODEL, true, KC, S, ALIGNED>), dim3((unsigned)nb), dim3(64), 0, st, a);
    else     hipLaunchKernelGGL((cpi_mean_dma_kernel<MODEL, false, KC, S, ALIGNED>), dim3((unsigned)nb), dim3(64), 0, st, a);
    return nb * 64;
}
template <int MODEL>
static long long launch_mean_dma(const MeanDmaCfg &c, bool avg, const PreArgs &a, hipStream_t st) {
#define CPI_DMA_CASE(K, S_, A_) if (c.kc == K && c.s == S_ && c.aligned == A_) return launch_mean_dma_one<MODEL, K, S_, (A_ != 0)>(avg, a, st);
    CPI_DMA_CASE(4, 2, 1) CPI_DMA_CASE(4, 2, 0) CPI_DMA_CASE(2, 3, 0) CPI_DMA_CASE(8, 1, 1)
    CPI_DMA_CASE(4, 1, 0) CPI_DMA_CASE(4, 1, 1) CPI_DMA_CASE(6, 1, 0) CPI_DMA_CASE(6, 2, 0)
#undef CPI_DMA_CASE
    return 0;
}
long long mean_dma(int model, const MeanDmaCfg &c, bool avg, const PreArgs &a, hipStream_t st) {
    return model == CPI_MODEL_V2 ? launch_mean_dma<2>(c, avg, a, st) : launch_mean_dma<1>(c, avg, a, st);
}
template <int MODEL, int L>
static void launch_mean_blk_L(bool avg, const PreArgs &a, hipStream_t st) {
    constexpr int WPB = 64 / L;
    const long long nb = (a.W + WPB - 1) / WPB;
    const size_t lds = ((size_t)WPB * (size_t)(a.N + 1) * 56 + 15) & ~(size_t)15;
    if (avg) hipLaunchKernelGGL((cpi_mean_blk_kernel<MODEL, true, L>), dim3((unsigned)nb), dim3(64), lds, st, a);
    else     hipLaunchKernelGGL((cpi_mean_blk_kernel<MODEL, false, L>), dim3((unsigned)nb), dim3(64), lds, st, a);
}
bool mean_blk(int model, int L, bool avg, const PreArgs &a, hipStream_t st) {
    const bool v2 = model == CPI_MODEL_V2;
    switch (L) {
        case 8: if (v2) launch_mean_blk_L<2, 8>(avg, a, st); else launch_mean_blk_L<1, 8>(avg, a, st); return true;
        case 16: if (v2) launch_mean_blk_L<2, 16>(avg, a, st); else launch_mean_blk_L<1, 16>(avg, a, st); return true;
        default: return false;
    }
}
// cpi_mean_line_kernel: the leading whole groups of 64 P windows of a dense batch; returns the number of windows handled
long long mean_line(int model, bool avg, const PreArgs &a, hipStream_t st) {
    if (a.first || a.count || a.update || a.tstart || a.N < 8 || a.N > 60000 || ((uintptr_t)a.knots & 127) != 0) return 0;
    const long long S = (long long)(a.N + 1) * 7;
    int P = 16;
    while (P > 1 && ((S * (P / 2)) & 15) == 0) P /= 2;            // smallest P with P S = 0 mod 16
    const long long groups = (a.W - 1) / (64ll * P);
    if (groups <= 0) return 0;
    const unsigned nb = (unsigned)(((groups + 7) / 8) * 8 * P);
    if (model == CPI_MODEL_V2) {
        if (avg) hipLaunchKernelGGL((cpi_mean_line_kernel<2, true>), dim3(nb), dim3(64), 0, st, a, P, groups);
        else     hipLaunchKernelGGL((cpi_mean_line_kernel<2, false>), dim3(nb), dim3(64), 0, st, a, P, groups);
    } else {
        if (avg) hipLaunchKernelGGL((cpi_mean_line_kernel<1, true>), dim3(nb), dim3(64), 0, st, a, P, groups);
        else     hipLaunchKernelGGL((cpi_mean_line_kernel<1, false>), dim3(nb), dim3(64), 0, st, a, P, groups);
    }
    return groups * 64 * P;
}
void tiled_fetch_probe(const TiledArgs &a, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL(cpi_tiled_fetch_probe_kernel, dim3((unsigned)((a.W + 63) / 64)), dim3(64), lds, st, a);
}
#endif

}  // namespace launch
}  // namespace cpi
