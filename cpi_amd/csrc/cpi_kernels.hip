// cpi_kernels.hip -- CDNA4 (gfx950) kernels + C-ABI of the batched continuous-preintegration engine.
//
// Kernels (one 64-lane wavefront per workgroup; all arithmetic f64 VALU, no MFMA -- the
// contractions are 3x3 / sparse 15x15):
//   cpi_mean_kernel<MODEL,JAC,AVG,L>   means (+ analytic bias Jacobians).  L lanes per window, each lane
//        integrates a contiguous run of intervals read through an LDS-staged, coalesced tile of IMU
//        knots, then an order-preserving shuffle tree composes the L segments.
//        Replaces CpiV1.h:67-259 / CpiV2.h:88-305.
//   cpi_cov_kernel<MODEL,AVG>          covariance (model 2: + compounded state transition -> Jacobians)
//        and means.  A 16- (model 1) or 32-lane (model 2) group owns one window; lane j owns column j
//        of P (or of Discrete_J_b).  Phase A computes the per-interval closed forms lane-parallel over
//        samples into LDS; phase C walks the samples sequentially: F x is lane-local, P F^T arrives
//        through a 9-row LDS transpose exchange; classic RK4.  Replaces CpiV1.h:266-353 / CpiV2.h:314-464.
//   cpi_factor_kernel<MODEL,WHITEN,LPF> evaluateError residual + dense 15x15 H1/H2; LPF (16/8/4) lanes per
//        factor, lane q emits columns q, q+LPF, ...  Replaces ImuFactorCPIv1.cpp:37-208 / ImuFactorCPIv2.cpp:38-212.
//   cpi_sqrt_info_kernel               R = chol_upper(P^-1) per factor (GTSAM noiseModel::Gaussian::Covariance, called
//        from ImuFactorCPIv1.h:82 / ImuFactorCPIv2.h:86): 16 lanes per factor, columns in registers, DPP row_share
//        broadcasts, triangular inverse fused into the factorisation.
//   cpi_predict_kernel<MODEL>          GraphSolver_IMU.cpp:263-307.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <algorithm>

#include "../../include/cpi_amd.h"
#include "../../include/cpi_amd_test.h"
#include "cpi_math.hpp"

using namespace cpi;


// ============================================================================================
// device helpers
// ============================================================================================
namespace {

__device__ __forceinline__ V3 ldv3(const double *p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ Q4 ldq4(const double *p) { Q4 q; q.x = p[0]; q.y = p[1]; q.z = p[2]; q.w = p[3]; return q; }
__device__ __forceinline__ M3 ldm3_cm(const double *p) {  // column-major 3x3
    M3 A;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) A.m[i][j] = p[j * 3 + i];
    return A;
}
__device__ __forceinline__ void stm3_cm(double *p, const M3 &A) {
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) p[j * 3 + i] = A.m[i][j];
}
__device__ __forceinline__ void stv3(double *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

__device__ __forceinline__ V3 shfl_down(V3 v, int d) {
    return mk(__shfl_down(v.x, d), __shfl_down(v.y, d), __shfl_down(v.z, d));
}
__device__ __forceinline__ M3 shfl_down(const M3 &A, int d) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = __shfl_down(A.m[i][j], d);
    return r;
}
__device__ __forceinline__ V3 shfl_down(V3 v, int d, int width) {
    return mk(__shfl_down(v.x, d, width), __shfl_down(v.y, d, width), __shfl_down(v.z, d, width));
}
__device__ __forceinline__ M3 shfl_up(const M3 &A, int d, int width) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = __shfl_up(A.m[i][j], d, width);
    return r;
}
// Shifts inside a 16-lane DPP row (= one model-1 window group of the covariance kernel): v_mov_b32_dpp row_shr / row_shl
// instead of ds_bpermute -- VALU moves with no LDS round trip to wait for.  Lanes whose source would lie outside the row
// keep their own value, exactly like __shfl_up / __shfl_down with width 16.  d is a constant after unrolling.
template <int CTRL>
__device__ __forceinline__ double dpp_mov64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ double row16_up(double v, int d) {      // lane j <- lane j - d
    switch (d) {
        case 1: return dpp_mov64<0x111>(v);
        case 2: return dpp_mov64<0x112>(v);
        case 4: return dpp_mov64<0x114>(v);
        case 8: return dpp_mov64<0x118>(v);
        default: return __shfl_up(v, d, 16);
    }
}
__device__ __forceinline__ double row16_down(double v, int d) {    // lane j <- lane j + d
    switch (d) {
        case 1: return dpp_mov64<0x101>(v);
        case 2: return dpp_mov64<0x102>(v);
        case 4: return dpp_mov64<0x104>(v);
        case 8: return dpp_mov64<0x108>(v);
        default: return __shfl_down(v, d, 16);
    }
}
__device__ __forceinline__ M3 row16_up_m3(const M3 &A, int d) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = row16_up(A.m[i][j], d);
    return r;
}
// Lane 15 of DPP rows 0 and 2 -> every lane of rows 1 and 3 (row_bcast:15, row_mask 0b1010); rows 0 and 2 keep their value.
__device__ __forceinline__ M3 row_bcast15_to_odd_rows(const M3 &A) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int lo = __double2loint(A.m[i][j]), hi = __double2hiint(A.m[i][j]);
            r.m[i][j] = __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x142, 0xa, 0xf, false),
                                         __builtin_amdgcn_update_dpp(lo, lo, 0x142, 0xa, 0xf, false));
        }
    return r;
}
template <int GROUP>
__device__ __forceinline__ M3 group_up(const M3 &A, int d) {
    if (GROUP != 16) return shfl_up(A, d, GROUP);
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = row16_up(A.m[i][j], d);
    return r;
}
template <int GROUP>
__device__ __forceinline__ V3 group_down(V3 v, int d) {
    if (GROUP != 16) return shfl_down(v, d, GROUP);
    return mk(row16_down(v.x, d), row16_down(v.y, d), row16_down(v.z, d));
}
template <int GROUP>
__device__ __forceinline__ double group_down(double v, int d) {
    return (GROUP != 16) ? __shfl_down(v, d, GROUP) : row16_down(v, d);
}
template <bool JAC>
__device__ __forceinline__ MeanState<JAC> shfl_down(const MeanState<JAC> &s, int d) {
    MeanState<JAC> r;
    r.R = shfl_down(s.R, d);
    r.alpha = shfl_down(s.alpha, d);
    r.beta = shfl_down(s.beta, d);
    r.DT = __shfl_down(s.DT, d);
    if (JAC) {
        r.Jq = shfl_down(s.Jq, d); r.Ja = shfl_down(s.Ja, d); r.Jb = shfl_down(s.Jb, d);
        r.Ha = shfl_down(s.Ha, d); r.Hb = shfl_down(s.Hb, d);
        r.Oa = s.Oa; r.Ob = s.Ob;
    }
    return r;
}
__device__ __forceinline__ GravAcc shfl_down(const GravAcc &g, int d) {
    GravAcc r;
    r.Gam = shfl_down(g.Gam, d); r.Lam = shfl_down(g.Lam, d);
    return r;
}
// Orders LDS traffic of a single-wavefront workgroup for the COMPILER only: a wave's DS instructions execute
// in issue order, so no counter drain (and no s_barrier) is needed between a write and a dependent read.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// lanes 4..7 of DPP rows 0 and 2 (= the clone lanes of the two 32-lane model-2 groups) <- lanes 0..3 of the same
// row; every other lane keeps its value.  row_shr:4, row_mask 0b0101, bank_mask 0b0010, bound_ctrl 0.
__device__ __forceinline__ double dpp_clone_shr4(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int nlo = __builtin_amdgcn_update_dpp(lo, lo, 0x114, 0x5, 0x2, false);
    const int nhi = __builtin_amdgcn_update_dpp(hi, hi, 0x114, 0x5, 0x2, false);
    return __hiloint2double(nhi, nlo);
}
// lanes 12..15 of every DPP row <- lanes 6..9 of the same row (row_shr:6, bank_mask 0b1000); all other lanes keep `old`.
__device__ __forceinline__ double dpp_shr6_bank3(double old, double src) {
    const int nlo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), 0x116, 0xf, 0x8, false);
    const int nhi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), 0x116, 0xf, 0x8, false);
    return __hiloint2double(nhi, nlo);
}
// Maximum over the wavefront, wave-uniform result.  DPP reduction (row_shr 1/2/4/8 -> lane 15 of each row holds the row
// maximum; row_bcast:15 / row_bcast:31 carry it across rows; lane 63 holds the total) instead of six dependent
// ds_bpermute round trips: it sits on the critical path of every wavefront's start-up.
__device__ __forceinline__ int wave_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}

struct PreArgs {
    long long W;
    int N;
    const double *knots;
    const long long *first;
    const int *count;
    const double *lin;
    const double *qk;
    double grav[3];
    double q4[4];      // sigma^2 of the four diagonal blocks of Q_c (CpiBase.h:54-57)
    int write_means;   // kernel writes DT/alpha/beta/q
    int write_jac;     // kernel writes the Jacobians it owns
    int dbg;           // development switches of the experimental kernels (0 in every shipped path)
    cpi_outputs out;
};

// ============================================================================================
// mean (+ analytic Jacobian) kernel
// ============================================================================================
#ifndef CPI_MEAN_WPS
#define CPI_MEAN_WPS 1
#endif
template <int MODEL, bool JAC, bool AVG, int L>
__global__ __launch_bounds__(64, (((MODEL == 2 && !JAC) || (MODEL == 1 && JAC)) && L == 1 ? 2 : CPI_MEAN_WPS)) void cpi_mean_kernel(PreArgs A) {
    constexpr int WPB = 64 / L;       // windows per wavefront
    // knots staged per lane per chunk: measured on MI355X -- 2 when a lane has several intervals (L <= 8; 20 k x 50 with
    // L = 3: 19.7 -> 18.4 us, 30 k with L = 2: 27.4 -> 24.8 us, 15 k with L = 4: 16.0 -> 15.3 us, 10 k with L = 6:
    // 12.5 -> 11.8 us once the padded second step of an odd last chunk is skipped), 1 when a wave is latency-bound
    // with few intervals per lane (L >= 12: 5 k windows 9.55 vs 9.65 us, 2.5 k 7.7 vs 8.0 us); 3 / 4 knots per chunk cost the second
    // wavefront per SIMD
    constexpr int C = (L <= 8 && !JAC) ? 2 : 1;
    constexpr int SEGD = 7 * C;       // doubles per lane per chunk
    constexpr int PITCH = SEGD + 1;   // odd pitch: conflict-free ds_read_b64 across the lanes of a half-wave
    __shared__ double tile[64 * PITCH];
    __shared__ unsigned long long segdesc[64];  // per lane-segment: (first double of the segment << 16) | intervals

    const int lane = threadIdx.x;
    const int grp = lane / L, l = lane - grp * L;
    long long w = (long long)blockIdx.x * WPB + grp;
    const bool valid = (w < A.W) && (grp < WPB);   // L not a power of two leaves 64 - WPB*L idle lanes
    if (grp >= WPB) w = (long long)blockIdx.x * WPB;   // idle lanes shadow the block's first window (stays near the block)
    if (w >= A.W) w = A.W - 1;
    const int n = A.count ? min(max(A.count[w], 0), A.N) : A.N;   // a count outside [0, N] must not corrupt the packed descriptors
    const long long k0 = A.first ? A.first[w] : w * (long long)(A.N + 1);
    const int per = (n + L - 1) / L;
    const int s0 = min(n, l * per), s1 = min(n, s0 + per);
    const int len = s1 - s0;
    const int maxlen = __builtin_amdgcn_readfirstlane(wave_max(len));   // wave-uniform: loop control stays scalar

    // (Deriving the descriptors of a dense layout arithmetically instead of through LDS was measured: +0.35 us per
    // 13 us launch -- the 64-bit integer arithmetic costs more than the shuffle reduction and the LDS round trip.)
    segdesc[lane] = ((unsigned long long)((k0 + s0) * 7) << 16) | (unsigned long long)(unsigned)len;

    const V3 bw = ldv3(A.lin + w * 6), ba = ldv3(A.lin + w * 6 + 3);
    V3 gk = mk(0, 0, 0);
    if (MODEL == 2) gk = mul(quat_2_Rot(ldq4(A.qk + w * 4)), mk(A.grav[0], A.grav[1], A.grav[2]));

    double pk[7];
    {
        const double *kb = A.knots + (k0 + s0) * 7;
#pragma unroll
        for (int i = 0; i < 7; i++) pk[i] = kb[i];  // knot s0 always exists (a window owns count+1 knots)
    }
    MeanState<JAC> st;
    mean_init(st);
    // model 2, mean-only, several lanes per window: a lane integrates its segment from the raw specific force and
    // accumulates the segment's gravity response (cpi_math.hpp: mean_step_v2seg); gravity is applied after the tree
    constexpr bool GSEG = (MODEL == 2) && !JAC && (L > 1);
    GravAcc ga;
    if (GSEG) grav_init(ga);
    __syncthreads();

    // Analytic-Jacobian variant of model 1, one lane per window (large batches): the recursion is bound by registers
    // (61 doubles of state + the per-interval 3x3 temporaries), not by HBM, so it streams its knots straight into
    // registers, one interval ahead, instead of through the coalescing LDS stage -- that frees the stage's address /
    // staging registers and lets two wavefronts share a SIMD (256 registers + 36 B of scratch each).  Measured inside
    // "V1 full" (covariance kernel + this one): 1.405 -> 1.376 ms per 100 k windows, 13.25 -> 13.10 ms per 1 M.  With
    // several lanes per window (small, latency-bound batches) it loses (10 k windows: 192 -> 205 us), so those keep the stage.
    constexpr bool DIRECT = JAC && (MODEL == 1) && (L == 1);
    if constexpr (DIRECT) {
        const double *kp = A.knots + (k0 + s0) * 7;
        double nx[7];
        {
            const double *kb = kp + 7 * min(1, len);
#pragma unroll
            for (int i = 0; i < 7; i++) nx[i] = kb[i];
        }
        for (int sidx = 0; sidx < maxlen; ++sidx) {
            double q[7];
#pragma unroll
            for (int i = 0; i < 7; i++) q[i] = nx[i];
            {
                const double *kb = kp + 7 * min(sidx + 2, len);   // knot s0 + len is the window segment's last: always valid
#pragma unroll
                for (int i = 0; i < 7; i++) nx[i] = kb[i];
            }
            mean_step<MODEL, JAC, AVG>(st, pk[0], q[0], mk(pk[1], pk[2], pk[3]), mk(pk[4], pk[5], pk[6]),
                                       mk(q[1], q[2], q[3]), mk(q[4], q[5], q[6]), bw, ba, gk, sidx < len);
#pragma unroll
            for (int i = 0; i < 7; i++) pk[i] = q[i];
        }
    } else {
    // Tile element idx = e*64 + lane belongs to segment idx / SEGD at offset idx % SEGD, so consecutive
    // lanes read consecutive doubles of (mostly) one segment: coalesced.  Everything that does not depend
    // on the chunk index is hoisted: per staged element a lane keeps one pointer and the last chunk for
    // which its knot exists (later chunks re-read that knot; the value is never consumed), so the hot loop
    // spends ~3 VALU per element on addressing and no load is ever out of bounds.
    double stage[SEGD];
    const double *sptr[SEGD];
    unsigned voff[SEGD];   // byte offset of the element from the block's first knot (dense layouts)
    int smax[SEGD];
    int tofs[SEGD];
    const double *blk0 = A.knots + (long long)blockIdx.x * WPB * (long long)(A.N + 1) * 7;   // wave-uniform
    {
        // (Issuing all SEGD descriptor reads before using the first -- one LDS round trip instead of SEGD dependent ones,
        // which hipcc keeps in program order with an s_waitcnt after each -- was measured: 12.55 vs 12.33 us per launch
        // at 10 k windows, i.e. slower; the wavefronts wait for the first HBM burst either way and start less staggered.)
        int seg = lane / SEGD, off = lane - seg * SEGD;
#pragma unroll
        for (int e = 0; e < SEGD; ++e) {
            const unsigned long long d = segdesc[seg];
            const long long base = (long long)(d >> 16);
            const int slen = (int)(d & 0xffffULL);
            const int kn = off / 7;                       // knot (1 + kn) of chunk 0
            const bool ok = slen >= 1 + kn;
            sptr[e] = A.knots + base + (ok ? 7 + off : off - 7 * kn);
            voff[e] = (unsigned)((sptr[e] - blk0) * 8);   // only used when safe_overread (then 0 <= offset < 2^32)
            smax[e] = ok ? (slen - 1 - kn) / C : 0;       // never-valid elements keep re-reading knot 0
            tofs[e] = seg * PITCH + off;
            off += 64 % SEGD; seg += 64 / SEGD;   // idx advances by 64 per staged element
            if (off >= SEGD) { off -= SEGD; seg += 1; }
        }
    }
    // Dense layout, not one of the last waves: reading a few knots past a short segment's end stays inside
    // the knot array, so every chunk is "block base + chunk stride (scalar) + constant lane offset".
    // (Not with per-window counts: the knots behind a short window's last interval belong to the caller's dense array and
    // may never have been written -- a NaN there would reach the state through 0 * NaN on the inactive steps.  The
    // per-element path below stops at the segment's end and re-reads its last, valid knot instead.)
    const bool safe_overread = (A.first == nullptr) && (A.count == nullptr) && ((long long)(blockIdx.x + 1) * WPB + 2 < A.W);
    auto issue = [&](int it) {
        if (safe_overread) {
            // scalar base (advanced by SALU) + constant 32-bit lane offsets: no vector arithmetic per element
            const char *cb = reinterpret_cast<const char *>(blk0) + (long long)it * (SEGD * 8);
#pragma unroll
            for (int e = 0; e < SEGD; ++e) {
                asm volatile("" : "+v"(voff[e]));   // keeps the zero-extension next to the load: `global_load v, v_off32, s[base]`
                stage[e] = *reinterpret_cast<const double *>(cb + voff[e]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < SEGD; ++e) { stage[e] = *sptr[e]; sptr[e] += (it < smax[e]) ? SEGD : 0; }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int e = 0; e < SEGD; ++e) tile[tofs[e]] = stage[e];
    };

    // One chunk ahead: the HBM round trip of chunk it+1 overlaps the FP64 work of chunk it.  Measured alternatives:
    // a TRUE two-chunk pipeline (two register stages, every path issuing the same loads so that hipcc emits the partial
    // wait s_waitcnt vmcnt(14) -- one conditional issue in the loop and it drains the queue with vmcnt(0)) is 8 % slower
    // at 10 k windows x 50 (13.5 vs 12.5 us: the first chunk's data queues behind the second's) and 5 % slower at 1 M;
    // a double-buffered LDS tile with the next chunk read back into registers during the integration: +2 %.
    // Per-wavefront time stamps explain why: with 1000 wavefronts in flight a chunk is 3.6 MB and takes 0.89 us
    // (0.74 us with 625 wavefronts, 1.2 us with 2000) -- the loop streams at ~4 TB/s and is paced by the memory
    // system, not by the latency of one wavefront's accesses.
    const int nchunks = (maxlen + C - 1) / C;
    if (nchunks > 0) issue(0);
    for (int it = 0; it < nchunks; ++it) {
        commit();
        __syncthreads();
        if (it + 1 < nchunks) issue(it + 1);
#pragma unroll   // C <= 2: the two steps of a chunk share one basic block (no knot copy between them)
        for (int c = 0; c < C; ++c) {
            const int s = it * C + c;
            if (C > 1 && s >= maxlen) break;   // wave-uniform: no lane has this interval (odd longest segment)
            const double *nk = &tile[lane * PITCH + c * 7];
            double q[7];
#pragma unroll
            for (int i = 0; i < 7; i++) q[i] = nk[i];
            if constexpr (GSEG)
                mean_step_v2seg<AVG>(st, ga, pk[0], q[0], mk(pk[1], pk[2], pk[3]), mk(pk[4], pk[5], pk[6]),
                                     mk(q[1], q[2], q[3]), mk(q[4], q[5], q[6]), bw, ba, s < len);
            else
                mean_step<MODEL, JAC, AVG>(st, pk[0], q[0], mk(pk[1], pk[2], pk[3]), mk(pk[4], pk[5], pk[6]),
                                           mk(q[1], q[2], q[3]), mk(q[4], q[5], q[6]), bw, ba, gk, s < len);
#pragma unroll
            for (int i = 0; i < 7; i++) pk[i] = q[i];
        }
        __syncthreads();
    }

    }   // !DIRECT

    // order-preserving composition tree over the L lanes of a window (earlier = lower lane)
#pragma unroll
    for (int stp = 1; stp < L; stp <<= 1) {
        MeanState<JAC> B = shfl_down(st, stp);
        GravAcc gB;
        if constexpr (GSEG) gB = shfl_down(ga, stp);
        if ((L & (L - 1)) != 0) {
            // L not a power of two: lane l + stp may belong to the next window -- compose with the identity instead
            if (l + stp >= L) { mean_init(B); if (GSEG) grav_init(gB); }
        }
        if constexpr (GSEG) grav_combine(ga, st, gB, B);   // needs st.R / B.DT before they are composed
        mean_combine(st, B);
    }
    if constexpr (GSEG) grav_apply(st, ga, gk);

    if (valid && l == 0) {
        if (A.write_means) {
            if (A.out.DT) A.out.DT[w] = st.DT;
            if (A.out.alpha) stv3(A.out.alpha + w * 3, st.alpha);
            if (A.out.beta) stv3(A.out.beta + w * 3, st.beta);
            if (A.out.q) {
                const Q4 q = rot_2_quat(st.R);
                double *p = A.out.q + w * 4;
                p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w;
            }
        }
        if (JAC && A.write_jac) {
            if (A.out.J_q) stm3_cm(A.out.J_q + w * 9, st.Jq);
            if (A.out.J_a) stm3_cm(A.out.J_a + w * 9, st.Ja);
            if (A.out.J_b) stm3_cm(A.out.J_b + w * 9, st.Jb);
            if (A.out.H_a) stm3_cm(A.out.H_a + w * 9, st.Ha);
            if (A.out.H_b) stm3_cm(A.out.H_b + w * 9, st.Hb);
            if (MODEL == 2) {
                if (A.out.O_a) stm3_cm(A.out.O_a + w * 9, st.Oa);
                if (A.out.O_b) stm3_cm(A.out.O_b + w * 9, st.Ob);
            }
        }
    }
}

// ============================================================================================
// mean kernel, large batches: knots streamed into LDS by the DMA path (global_load_lds_dwordx4)
// ============================================================================================
// One lane per window, 64 consecutive windows per wavefront (dense layout).  A STAGE is KC knots of every window of
// the wavefront: 64 x KC x 56 B.  Stages land in a ring of S LDS slots through LDS-DMA loads, i.e. without passing
// through (and without costing) vector registers: the cpi_mean_kernel stage of 14 doubles + 14 pointers + 42 address
// words per lane is gone, the prefetch distance is S - 1 whole stages, and a window contributes KC x 56 contiguous
// bytes per request instead of 112 (DRAM page locality: a pure-read kernel with this access pattern streams 2.86 GB in
// 0.53 ms with 112-byte pieces, 0.48 ms with 448-byte pieces, 0.44 ms linearly -- DESIGN.md 3.1).
//
// LDS-DMA writes "wave-uniform base (M0) + 16 x lane", so the LDS image of one DMA instruction is fixed: lane l's 16
// bytes at 16 l.  Lane l of instruction j fetches piece (l mod PPW) of window j*WPI + l / PPW of the block: the PPW
// lanes of a window read PPW x 16 contiguous bytes (coalesced), WPI = 64 / PPW windows per instruction, 64 mod PPW
// idle lanes re-fetch a valid address.  ALIGNED: every piece is fetched from a 16-byte aligned address -- a window
// starts on an 8-byte boundary only (56-byte knots), so a stage is fetched as the aligned superset of PPW = KC*3.5 + 1
// pieces and the reader skips its window's leading 0 / 8 bytes.  The odd piece count also spreads the readers' rows over
// the LDS banks (row pitch 240 B -> 2-way conflicts on the 8-byte reads; 224 B would be 4-way, 256 B 32-way).
// The global address of instruction j is "scalar base (SALU) + constant 32-bit lane offset": no vector address
// arithmetic at all.  Ordering: the wave's own counted s_waitcnt vmcnt is what orders its ds_reads behind its LDS-DMA
// (MI355X_MICROARCH.md item 7; single-wave workgroup, no barrier needed); a slot is re-armed only after an
// lgkmcnt(0) has retired the reads of its previous contents.
// MEASURED (MI355X, profiles/r02_mean_lds_dma.md; 1 M x 50, cpi_mean_kernel = 0.669 ms): KC,S = 4,2 unaligned 0.675 ms,
// 4,2 aligned 0.685, 2,3 0.72, 4,3 0.71, 8,1 0.75, 8,2 (2 wavefronts per CU) 1.03; 100 k x 50: 0.119 vs 0.073 ms (5
// wavefronts per CU by LDS = two rounds of wavefronts instead of one).  The staging registers are gone (110 VGPRs instead
// of 212) and 16 KB per wavefront are in flight, yet nothing is gained: the strided multi-stream pattern itself delivers
// ~4.8 TB/s of DMA traffic.  NOT the default: reachable through CPI_AMD_MEAN_DMA=KC,S,A for A/B measurements only.
template <int KC, bool ALIGNED>
struct DmaGeom {
    static constexpr int PPW = (KC * 56) / 16 + (ALIGNED ? 1 : 0);   // 16-byte pieces per window and stage
    static constexpr int WPI = 64 / PPW;                             // windows per DMA instruction
    static constexpr int NI = (64 + WPI - 1) / WPI;                  // DMA instructions per stage
    static constexpr int SLOT = NI * 1024;                           // bytes per ring slot
    static_assert((KC * 56) % 16 == 0, "a stage is a whole number of 16-byte pieces (KC even)");
    static_assert(!ALIGNED || (WPI % 2 == 0), "alignment phase of an instruction's first window must not depend on j");
};
__device__ __forceinline__ void glds16(unsigned voff, const void *sbase, unsigned lds_dst) {
    unsigned keep;
    // M0 carries the LDS destination; hipcc does not preserve it around a statement, so it is set and restored here
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int N_> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N_) : "memory"); }
template <int MODEL, bool AVG, int KC, int S, bool ALIGNED>
__global__ __launch_bounds__(64, 1) void cpi_mean_dma_kernel(PreArgs A) {
    typedef DmaGeom<KC, ALIGNED> G;
    static_assert((S - 1) * G::NI <= 63, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(1024))) char ring[S * G::SLOT];
    const int lane = threadIdx.x;
    const long long w = (long long)blockIdx.x * 64 + lane;          // every window of a block exists (host guarantees)
    const int n = A.count ? min(max(A.count[w], 0), A.N) : A.N;
    const int nmax = __builtin_amdgcn_readfirstlane(wave_max(n));
    const long long wstride = (long long)(A.N + 1) * 56;            // bytes per window
    const char *blk = reinterpret_cast<const char *>(A.knots) + (long long)blockIdx.x * 64 * wstride;   // wave-uniform

    // ---- DMA role of this lane (constant over instructions and stages)
    const int dw = lane / G::PPW, dp = lane - dw * G::PPW;          // window within the instruction, piece within the window
    const bool idle = dw >= G::WPI;
    // alignment phase: byte address of knot 1 of window (instr j, dw) = blk + (j WPI + dw) wstride + 56 + stage offset;
    // KC and WPI even -> its bit 3 depends on dw alone
    unsigned voff, rshift = 0;
    {
        const unsigned long long a0 = (unsigned long long)(blk + 56);
        const unsigned ph = ALIGNED ? (unsigned)((a0 + (unsigned long long)(idle ? 0 : dw) * (unsigned long long)wstride) & 8ull) : 0u;
        // scalar base is biased by -16 so that the lane offset stays non-negative
        voff = (unsigned)((idle ? 0 : dw) * wstride) + 16u * (unsigned)(idle ? 0 : dp) + 16u - ph;
        if (ALIGNED) {
            const int rw = lane % G::WPI;                            // this lane's OWN window sits at row rw of instruction lane / WPI
            rshift = (unsigned)((a0 + (unsigned long long)rw * (unsigned long long)wstride) & 8ull);
        }
    }
    const unsigned ring_base = (unsigned)(size_t)((__attribute__((address_space(3))) char *)ring);
    const int rd_off = (lane / G::WPI) * 1024 + (lane % G::WPI) * (G::PPW * 16) + (int)rshift;   // reader: own window's row

    const int nst = (nmax + KC - 1) / KC;
    const int dbg = A.dbg;   // development: 1 = no arithmetic, 2 = no fetch
    auto issue = [&](int st) {
        if (dbg & 2) return;
        const char *sb = blk + 56 - 16 + (long long)st * (KC * 56);
        const unsigned dst = ring_base + (unsigned)(st % S) * G::SLOT;
#pragma unroll
        for (int j = 0; j < G::NI; ++j) glds16(voff, sb + (long long)j * G::WPI * wstride, dst + j * 1024);
    };

    // prologue: S - 1 stages in flight, then this lane's first knot and linearisation point through ordinary loads
#pragma unroll
    for (int p = 0; p < S - 1; ++p) if (p < nst) issue(p);
    V3 bw = ldv3(A.lin + w * 6), ba = ldv3(A.lin + w * 6 + 3);
    V3 gk = mk(0, 0, 0);
    if (MODEL == 2) gk = mul(quat_2_Rot(ldq4(A.qk + w * 4)), mk(A.grav[0], A.grav[1], A.grav[2]));
    double pk[7];
    {
        const double *kb = A.knots + w * (long long)(A.N + 1) * 7;
#pragma unroll
        for (int i = 0; i < 7; i++) pk[i] = kb[i];
    }
    MeanState<false> st_;
    mean_init(st_);
    // The ordinary loads above must be complete BEFORE the loop: hipcc would otherwise wait for them at their first
    // use inside it -- an s_waitcnt vmcnt(0) executed in every iteration, which also drains the prefetched stages
    // (its counter model does not include the LDS-DMA instructions).
#pragma unroll
    for (int i = 0; i < 7; i++) asm volatile("" : "+v"(pk[i]));
    asm volatile("" : "+v"(bw.x), "+v"(bw.y), "+v"(bw.z), "+v"(ba.x), "+v"(ba.y), "+v"(ba.z));
    asm volatile("" : "+v"(gk.x), "+v"(gk.y), "+v"(gk.z));

    for (int st = 0; st < nst; ++st) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // reads of the slot about to be re-armed have retired
        const int ahead = min(S - 1, nst - 1 - st);                  // stages younger than st that are (or get) in flight
        if (S > 1 && st + S - 1 < nst) issue(st + S - 1);
        if (S == 1) issue(st);
        // stage st has landed once at most `ahead` stages' worth of younger DMA instructions are outstanding
        if (S == 1 || ahead == 0) wait_vmcnt<0>();
        else if (ahead == 1) wait_vmcnt<(S > 1 ? 1 : 0) * G::NI>();
        else if (ahead == 2) wait_vmcnt<(S > 2 ? 2 : 0) * G::NI>();
        else wait_vmcnt<(S > 3 ? 3 : 0) * G::NI>();
        const char *slot = ring + (st % S) * G::SLOT + rd_off;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int s = st * KC + c;
            if (s >= nmax || (dbg & 1)) break;                       // wave-uniform
            const double *nk = reinterpret_cast<const double *>(slot + 56 * c);
            double q[7];
#pragma unroll
            for (int i = 0; i < 7; i++) q[i] = nk[i];
            mean_step<MODEL, false, AVG>(st_, pk[0], q[0], mk(pk[1], pk[2], pk[3]), mk(pk[4], pk[5], pk[6]),
                                         mk(q[1], q[2], q[3]), mk(q[4], q[5], q[6]), bw, ba, gk, s < n);
#pragma unroll
            for (int i = 0; i < 7; i++) pk[i] = q[i];
        }
    }

    if (A.out.DT) A.out.DT[w] = st_.DT;
    if (A.out.alpha) stv3(A.out.alpha + w * 3, st_.alpha);
    if (A.out.beta) stv3(A.out.beta + w * 3, st_.beta);
    if (A.out.q) {
        const Q4 q = rot_2_quat(st_.R);
        double *p = A.out.q + w * 4;
        p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w;
    }
}

// ============================================================================================
// mean kernel on the TILED layout: knots of 64 windows interleaved per step (cpi_preintegrate_tiled_batch)
// ============================================================================================
// tiles[b][s][k][i] = field k (t, w, a) of knot s of window 64 b + i.  A wavefront owns tile b, lane i window 64 b + i, and
// step s reads seven fully coalesced 512-byte rows: the whole batch is ONE linear stream per wavefront, every byte
// fetched once, no LDS, no staging, ~100 registers (4 wavefronts per SIMD) -- the layout the recursion wants on this
// memory system, for producers that can write it (a batch assembler that places knot s of window w at its tile slot instead of
// at w (N+1) + s costs nothing extra).  Knots are prefetched three steps ahead in registers.
struct TiledArgs {
    long long W;
    int N;
    const double *tiles;
    const int *count;
    const double *lin;
    const double *qk;
    double grav[3];
    cpi_outputs out;
    int dbg;   // measurement only (CPI_AMD_BLK_MODE): 1 = fetch without arithmetic
    long long ts, ss;   // doubles between consecutive tiles / consecutive steps of a tile
};
#ifndef CPI_TILED_OCC
#define CPI_TILED_OCC (MODEL == 2 ? 2 : 3)
#endif
#ifndef CPI_TILED_BUFS
#define CPI_TILED_BUFS 5
#endif
// SPLIT (small batches: fewer tiles than the chip has SIMDs): a workgroup of S = blockDim.x / 64 wavefronts owns the
// tile; wavefront j integrates the steps [j per, (j + 1) per) of all 64 windows from the identity (model 2: from the raw
// specific force, with the segment's gravity response -- cpi_math.hpp mean_step_v2seg), parks its segment in LDS, and
// wavefront 0 composes the S segments in order (mean_combine / grav_combine: the composition cpi_mean_kernel uses
// across the lanes of a window).  Each wavefront still reads one linear stream.
template <int MODEL, bool AVG, bool COUNTED, bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 512 : 64, SPLIT ? 1 : CPI_TILED_OCC) void cpi_mean_tiled_kernel(TiledArgs A) {
    constexpr bool GSEG = SPLIT && MODEL == 2;
    constexpr int NF = GSEG ? 34 : 16;            // doubles of a parked segment
    extern __shared__ double seg[];               // [S - 1][NF][64]
    const int lane = threadIdx.x & 63;
    const int j = SPLIT ? (int)(threadIdx.x >> 6) : 0, S = SPLIT ? (int)(blockDim.x >> 6) : 1;
    const long long w = (long long)blockIdx.x * 64 + lane;
    const bool valid = w < A.W;
    const long long wc = valid ? w : A.W - 1;
    const int n = valid ? (COUNTED ? min(max(A.count[wc], 0), A.N) : A.N) : 0;
    const int nmax = COUNTED ? __builtin_amdgcn_readfirstlane(wave_max(n)) : A.N;
    const int per = SPLIT ? (A.N + S - 1) / S : A.N;
    const int sb = __builtin_amdgcn_readfirstlane(j * per), se = min(sb + per, nmax);   // this wavefront's steps
    const double *tb = A.tiles + (long long)blockIdx.x * A.ts + lane;   // a step of a tile: 448 doubles = 7 fields x 64 windows
    const V3 bw = ldv3(A.lin + wc * 6), ba = ldv3(A.lin + wc * 6 + 3);
    V3 gk = mk(0, 0, 0);
    if (MODEL == 2 && j == 0) gk = mul(quat_2_Rot(ldq4(A.qk + wc * 4)), mk(A.grav[0], A.grav[1], A.grav[2]));
    auto load = [&](double (&k)[7], int s) {
        // COUNTED: past its own last knot a lane re-reads that knot (dt = 0) -- what lies behind it in the column is
        // never read.  Otherwise the row offset is wave-uniform (scalar address arithmetic).
        const double *p = tb + (long long)(COUNTED ? min(s, n) : min(s, A.N)) * A.ss;
#pragma unroll
        for (int f = 0; f < 7; f++) k[f] = p[f * 64];
    };
    MeanState<false> st;
    mean_init(st);
    GravAcc ga;
    if (GSEG) grav_init(ga);
    // the knot buffers rotate by NAME over one unrolled trip (a rolled loop spends 28 v_mov_b64 per step on it)
#define CPI_TSTEP(a, b, e, S_)                                                                                        \
    load(e, (S_) + CPI_TILED_BUFS - 1);                                                                               \
    if constexpr (GSEG)                                                                                               \
        mean_step_v2seg<AVG>(st, ga, a[0], b[0], mk(a[1], a[2], a[3]), mk(a[4], a[5], a[6]), mk(b[1], b[2], b[3]),    \
                             mk(b[4], b[5], b[6]), bw, ba, (S_) < n);                                                 \
    else                                                                                                              \
        mean_step<MODEL, false, AVG>(st, a[0], b[0], mk(a[1], a[2], a[3]), mk(a[4], a[5], a[6]), mk(b[1], b[2], b[3]), \
                                     mk(b[4], b[5], b[6]), bw, ba, gk, (S_) < n)
#if CPI_TILED_BUFS == 5
    double k0[7], k1[7], k2[7], k3[7], k4[7];
    load(k0, sb); load(k1, sb + 1); load(k2, sb + 2); load(k3, sb + 3);
    for (int s = sb; s < se; s += 5) {
        CPI_TSTEP(k0, k1, k4, s);
        if (s + 1 >= se) break;
        CPI_TSTEP(k1, k2, k0, s + 1);
        if (s + 2 >= se) break;
        CPI_TSTEP(k2, k3, k1, s + 2);
        if (s + 3 >= se) break;
        CPI_TSTEP(k3, k4, k2, s + 3);
        if (s + 4 >= se) break;
        CPI_TSTEP(k4, k0, k3, s + 4);
    }
#elif CPI_TILED_BUFS == 4
    double k0[7], k1[7], k2[7], k3[7];
    load(k0, sb); load(k1, sb + 1); load(k2, sb + 2);
    for (int s = sb; s < se; s += 4) {
        CPI_TSTEP(k0, k1, k3, s);
        if (s + 1 >= se) break;
        CPI_TSTEP(k1, k2, k0, s + 1);
        if (s + 2 >= se) break;
        CPI_TSTEP(k2, k3, k1, s + 2);
        if (s + 3 >= se) break;
        CPI_TSTEP(k3, k0, k2, s + 3);
    }
#else
    double k0[7], k1[7], k2[7];
    load(k0, sb); load(k1, sb + 1);
    for (int s = sb; s < se; s += 3) {
        CPI_TSTEP(k0, k1, k2, s);
        if (s + 1 >= se) break;
        CPI_TSTEP(k1, k2, k0, s + 1);
        if (s + 2 >= se) break;
        CPI_TSTEP(k2, k0, k1, s + 2);
    }
#endif
#undef CPI_TSTEP
    if constexpr (SPLIT) {
        auto park = [&](int f, double v) { seg[((j - 1) * NF + f) * 64 + lane] = v; };
        if (j > 0) {
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) park(r * 3 + c, st.R.m[r][c]);
            park(9, st.alpha.x); park(10, st.alpha.y); park(11, st.alpha.z);
            park(12, st.beta.x); park(13, st.beta.y); park(14, st.beta.z); park(15, st.DT);
            if constexpr (GSEG) {
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int c = 0; c < 3; c++) { park(16 + r * 3 + c, ga.Gam.m[r][c]); park(25 + r * 3 + c, ga.Lam.m[r][c]); }
            }
        }
        __syncthreads();
        if (j > 0) return;
        for (int jj = 1; jj < S; ++jj) {        // earlier o later, in order
            const double *sp = seg + (jj - 1) * NF * 64 + lane;
            MeanState<false> B;
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) B.R.m[r][c] = sp[(r * 3 + c) * 64];
            B.alpha = mk(sp[9 * 64], sp[10 * 64], sp[11 * 64]);
            B.beta = mk(sp[12 * 64], sp[13 * 64], sp[14 * 64]);
            B.DT = sp[15 * 64];
            if constexpr (GSEG) {
                GravAcc gB;
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int c = 0; c < 3; c++) { gB.Gam.m[r][c] = sp[(16 + r * 3 + c) * 64]; gB.Lam.m[r][c] = sp[(25 + r * 3 + c) * 64]; }
                grav_combine(ga, st, gB, B);    // before mean_combine: needs st.R and B.DT as they are
            }
            mean_combine(st, B);
        }
        if constexpr (GSEG) grav_apply(st, ga, gk);
    }
    if (!valid) return;
    if (A.out.DT) A.out.DT[w] = st.DT;
    if (A.out.alpha) stv3(A.out.alpha + w * 3, st.alpha);
    if (A.out.beta) stv3(A.out.beta + w * 3, st.beta);
    if (A.out.q) {
        const Q4 q = rot_2_quat(st.R);
        double *p = A.out.q + w * 4;
        p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w;
    }
}
// measurement only (CPI_AMD_BLK_MODE=1): the tiled stream alone -- the same loads, one add per value
__global__ __launch_bounds__(64, 3) void cpi_tiled_fetch_probe_kernel(TiledArgs A) {
    const int lane = threadIdx.x;
    const long long w = (long long)blockIdx.x * 64 + lane;
    const double *tb = A.tiles + (long long)blockIdx.x * A.ts + lane;
    auto load = [&](double (&k)[7], int s) {
        const double *p = tb + (long long)min(s, A.N) * A.ss;
#pragma unroll
        for (int f = 0; f < 7; f++) k[f] = p[f * 64];
    };
    double k0[7], k1[7], k2[7], k3[7], acc = 0;
    load(k0, 0); load(k1, 1); load(k2, 2); load(k3, 3);
    for (int s = 0; s < A.N; ++s) {
        double k4[7];
        load(k4, s + 4);
#pragma unroll
        for (int f = 0; f < 7; f++) { acc += k0[f]; k0[f] = k1[f]; k1[f] = k2[f]; k2[f] = k3[f]; k3[f] = k4[f]; }
    }
    if (w < A.W && A.out.DT) A.out.DT[w] = acc;
}
// dense knots[W][N+1][7] -> tiles[ceil(W/64)][N+1][7][64] (windows past W replicate window W - 1: finite padding)
__global__ __launch_bounds__(256) void cpi_tile_knots_kernel(long long W, int N, const double *knots, double *tiles, long long ts, long long ss) {
    const long long total = ((W + 63) / 64) * (long long)(N + 1) * 448;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
        const int i = (int)(o & 63);
        const long long r = o >> 6;
        const int f = (int)(r % 7);
        const long long bs = r / 7;
        const int sidx = (int)(bs % (N + 1));
        const long long b = bs / (N + 1);
        const long long w = min(b * 64 + i, W - 1);
        tiles[b * ts + sidx * ss + f * 64 + i] = knots[(w * (N + 1) + sidx) * 7 + f];
    }
}

// ============================================================================================
// mean kernel, block-resident: a wavefront owns 64 / L CONSECUTIVE WHOLE windows (dense layout)
// ============================================================================================
// The wavefront's windows are one contiguous byte range of the knot array (64/L x (N+1) x 56 B -- 22.8 KB for
// L = 8, N = 50).  It is fetched LINEARLY by LDS-DMA, 1 KiB per instruction, every 128-byte line exactly once, and
// lands in LDS as the exact memory image; nothing passes through registers and there is no per-chunk dependency on
// the memory system (cpi_mean_kernel walks 9 serial chunks at 10 k windows).  Lane l of a window then integrates its
// contiguous run of ceil(n / L) intervals straight out of LDS and the L segments are composed by the order-preserving
// tree of cpi_mean_kernel (DPP row shifts: no LDS crossbar).  LDS per wavefront is dynamic (= the block's bytes):
// 7 wavefronts share a CU at N = 50 and overlap each other's fetch and arithmetic.
// MEASURED (MI355X, 1 M x 50, profiles/r02_mean_lds_dma.md): the fetch alone runs at 6.1 TB/s (0.48 ms), the arithmetic
// alone takes 0.53 ms (7 intervals + 3 tree levels per lane: ~45 % more FP64 instructions than one lane per window),
// together 0.675 ms -- the same as cpi_mean_kernel (0.67 ms), whose strided pattern is slower to fetch but whose
// arithmetic is minimal.  A persistent variant (buffer re-armed under the tree) was slower (0.72 ms).  NOT the default:
// reachable through CPI_AMD_MEAN_BLK=L for A/B measurements only.
// zero-filling row shift (lanes whose source lies outside the 16-lane row read 0): one v_mov_b32_dpp per half, no copy
template <int CTRL>
__device__ __forceinline__ double dpp_mov64z(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true),
                            __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ double row16_down0(double v, int d) {    // lane j <- lane j + d, 0 past the end of the row
    switch (d) {
        case 1: return dpp_mov64z<0x101>(v);
        case 2: return dpp_mov64z<0x102>(v);
        case 4: return dpp_mov64z<0x104>(v);
        default: return dpp_mov64z<0x108>(v);
    }
}
template <int MODEL, bool AVG, int L>
__global__ __launch_bounds__(64, 1) void cpi_mean_blk_kernel(PreArgs A) {
    static_assert((L & (L - 1)) == 0 && L >= 2 && L <= 32, "L lanes per window, power of two");
    constexpr int WPB = 64 / L;
    extern __shared__ __attribute__((aligned(1024))) char blkmem[];
    const int lane = threadIdx.x;
    const int grp = lane / L, l = lane % L;
    const long long w0 = (long long)blockIdx.x * WPB;
    const bool valid = (w0 + grp) < A.W;
    const long long w = valid ? w0 + grp : A.W - 1;
    const int nwin = (int)min((long long)WPB, A.W - w0);            // windows of this block (wave-uniform)
    const int wstride = (A.N + 1) * 56;                             // bytes per window (host guarantees WPB * wstride <= 64 KB)
    const int total = nwin * wstride;
    const char *blk = reinterpret_cast<const char *>(A.knots) + w0 * (long long)wstride;
    const unsigned ring_base = (unsigned)(size_t)((__attribute__((address_space(3))) char *)blkmem);
    const int dbg = A.dbg;   // development: 1 = no arithmetic, 2 = no fetch (where the time of this design goes)

    // ---- linear fetch of the block: instruction j moves bytes [1024 j, 1024 j + 1024)
    if (!(dbg & 2)) {
        const int nfull = total >> 10;
        const unsigned v16 = 16u * (unsigned)lane;
        for (int j = 0; j < nfull; ++j) glds16(v16, blk + ((long long)j << 10), ring_base + ((unsigned)j << 10));
        // the last, partial instruction runs with the lanes past the end of the block masked off (no load, no LDS write)
        const unsigned off = (unsigned)(nfull << 10) + v16;
        if ((int)(off + 16u) <= total) glds16(off, blk, ring_base + ((unsigned)nfull << 10));
    }
    const int n = valid ? (A.count ? min(max(A.count[w], 0), A.N) : A.N) : 0;
    const int per = (n + L - 1) / L;
    const int s0 = min(n, l * per), s1 = min(n, s0 + per);
    const int len = s1 - s0;
    const int maxlen = (dbg & 1) ? 0 : __builtin_amdgcn_readfirstlane(wave_max(len));
    V3 bw = ldv3(A.lin + w * 6), ba = ldv3(A.lin + w * 6 + 3);
    V3 gk = mk(0, 0, 0);
    if (MODEL == 2) gk = mul(quat_2_Rot(ldq4(A.qk + w * 4)), mk(A.grav[0], A.grav[1], A.grav[2]));
    // a block of 8-byte-odd length ends in the middle of a 16-byte piece: its last double comes through a register
    double lastd = 0.0;
    const bool patch = (total & 15) != 0;
    if (patch) lastd = *reinterpret_cast<const double *>(blk + total - 8);
    asm volatile("" : "+v"(bw.x), "+v"(bw.y), "+v"(bw.z), "+v"(ba.x), "+v"(ba.y), "+v"(ba.z), "+v"(lastd));
    asm volatile("" : "+v"(gk.x), "+v"(gk.y), "+v"(gk.z));
    wait_vmcnt<0>();                                                // the wave's own counted wait orders its ds_reads behind its LDS-DMA
    if (patch && lane == 0) *reinterpret_cast<double *>(blkmem + total - 8) = lastd;
    wave_lds_fence();

    constexpr bool GSEG = (MODEL == 2);
    MeanState<false> st;
    mean_init(st);
    GravAcc ga;
    if (GSEG) grav_init(ga);
    const double *kp = reinterpret_cast<const double *>(blkmem + (valid ? grp : 0) * wstride) + (long long)s0 * 7;
    double pk[7];
#pragma unroll
    for (int i = 0; i < 7; i++) pk[i] = kp[i];
    for (int s = 0; s < maxlen; ++s) {
        const double *nk = kp + 7 * min(s + 1, len);                 // knot s0 + len is the segment's last: always inside the block
        double q[7];
#pragma unroll
        for (int i = 0; i < 7; i++) q[i] = nk[i];
        if constexpr (GSEG)
            mean_step_v2seg<AVG>(st, ga, pk[0], q[0], mk(pk[1], pk[2], pk[3]), mk(pk[4], pk[5], pk[6]),
                                 mk(q[1], q[2], q[3]), mk(q[4], q[5], q[6]), bw, ba, s < len);
        else
            mean_step<MODEL, false, AVG>(st, pk[0], q[0], mk(pk[1], pk[2], pk[3]), mk(pk[4], pk[5], pk[6]),
                                         mk(q[1], q[2], q[3]), mk(q[4], q[5], q[6]), bw, ba, gk, s < len);
#pragma unroll
        for (int i = 0; i < 7; i++) pk[i] = q[i];
    }
    // order-preserving composition over the L lanes of a window; partners sit inside one 16-lane DPP row (L <= 16) or
    // one row further (L = 32: one LDS shuffle level)
#pragma unroll
    for (int stp = 1; stp < L; stp <<= 1) {
        MeanState<false> B;
        GravAcc gB;
        if (stp < 16) {
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) B.R.m[i][j] = row16_down0(st.R.m[i][j], stp);
            B.alpha = mk(row16_down0(st.alpha.x, stp), row16_down0(st.alpha.y, stp), row16_down0(st.alpha.z, stp));
            B.beta = mk(row16_down0(st.beta.x, stp), row16_down0(st.beta.y, stp), row16_down0(st.beta.z, stp));
            B.DT = row16_down0(st.DT, stp);
            if constexpr (GSEG) {
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) { gB.Gam.m[i][j] = row16_down0(ga.Gam.m[i][j], stp); gB.Lam.m[i][j] = row16_down0(ga.Lam.m[i][j], stp); }
            }
        } else {
            B = shfl_down(st, stp);
            if constexpr (GSEG) gB = shfl_down(ga, stp);
        }
        if constexpr (GSEG) grav_combine(ga, st, gB, B);
        mean_combine(st, B);
    }
    if constexpr (GSEG) grav_apply(st, ga, gk);
    if (valid && l == 0) {
        if (A.out.DT) A.out.DT[w] = st.DT;
        if (A.out.alpha) stv3(A.out.alpha + w * 3, st.alpha);
        if (A.out.beta) stv3(A.out.beta + w * 3, st.beta);
        if (A.out.q) {
            const Q4 q = rot_2_quat(st.R);
            double *p = A.out.q + w * 4;
            p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w;
        }
    }
}

// ============================================================================================
// covariance (+ state transition) kernel
// ============================================================================================
// Measured on MI355X: forcing two co-resident wavefronts per SIMD (<= 256 registers) costs spills and does not pay;
// the recursion runs one wavefront per SIMD and hides LDS latency with instruction-level parallelism instead.
// Two co-resident wavefronts per SIMD hide the LDS exchange latency of the recursion: <= 256 registers and
// <= 20 KB of LDS per wavefront.  The latter is why a phase-A pass stages GROUP/2 intervals per window
// (half the lanes take part in it); measured on MI355X against the one-wave-per-SIMD variant:
// model 2 5.0 -> 3.8 ms, model 1 2.27 -> 2.0 ms per 100 k windows.
#ifndef CPI_COV_WPS
#define CPI_COV_WPS 2
#endif
template <int MODEL, bool AVG>
__global__ __launch_bounds__(64, CPI_COV_WPS) void cpi_cov_kernel(PreArgs A) {
    typedef CovDims<MODEL> D;
    constexpr int GROUP = D::GROUP;   // lanes per window
    constexpr int G = 64 / GROUP;     // windows per wavefront
    // intervals per window staged by one phase-A pass: as many as 20 KB of LDS per wavefront (two wavefronts per
    // SIMD) leave room for next to the bank-conflict-free exchange area -- 14 records pitched 26 doubles (model 1),
    // 23 pitched 42 (model 2).  A pass used to cost as much as 2.3-2.5 intervals of phase C (68 / 114 us per pass at
    // 100 k windows with ds_bpermute scans), so fewer passes matter: 50 samples = 4 passes (model 1), 3 (model 2).
    constexpr int CH = (MODEL == 1) ? 14 : 23;
    static_assert(CH <= GROUP, "one lane per staged interval");
    constexpr int EP = EXCH_PITCH;
    constexpr int IRD = IrPitch<MODEL>::V;
    __shared__ __attribute__((aligned(16))) double irs[G * CH * IRD];          // interval records (phase A -> C)
    __shared__ __attribute__((aligned(256))) double exch[exch_doubles(G)];   // transpose exchange (bank-conflict-free placement, cpi_math.hpp)
    __shared__ __attribute__((aligned(16))) double gsh[G * GS_DOUBLES];        // carried rotation / means per window

    const int lane = threadIdx.x;
    const int g = lane / GROUP, j = lane % GROUP;
    long long w = (long long)blockIdx.x * G + g;
    const bool valid = w < A.W;
    if (!valid) w = A.W - 1;
    const int n = A.count ? min(max(A.count[w], 0), A.N) : A.N;
    const long long k0 = A.first ? A.first[w] : w * (long long)(A.N + 1);
    const int nmax = wave_max(n);

    const double q4[4] = { A.q4[0], A.q4[1], A.q4[2], A.q4[3] };

    const int jj = cov_col_of_lane<MODEL>(j);  // column owned by this lane; idle lanes (NCOL) run as a harmless zero transition column
    CovLane<MODEL> Ln;
    cov_init(Ln, jj, q4);
    double *ex_g = exch + g * EXCH_WIN;
    // every row starts on a 16-B boundary; said explicitly, or the row reads degrade from ds_read_b128 to ds_read_b64
    const double *ex_row = exch + (cov_row_off<MODEL>(G, g, jj) & ~1);   // (the offset is even; the mask lets the compiler see it)
    const int hoff = cov_h_offset<MODEL>(jj);
    double *gs = gsh + g * GS_DOUBLES;
    for (int i = lane; i < exch_doubles(G); i += 64) exch[i] = 0.0;
    if (j == 0) {
        cov_gs_init(gs);
        if (MODEL == 2) put3(gs + GS_GK, mul(quat_2_Rot(ldq4(A.qk + w * 4)), mk(A.grav[0], A.grav[1], A.grav[2])));
    }
    __syncthreads();
    cov_exch_init<MODEL>(exch, G, jj, q4);

    for (int base = 0; base < nmax; base += CH) {
        // ---- phase A: lane (g, j) owns interval base + j of its window: closed forms, then the running rotation
        // at its start by a prefix product over the group's lanes, then everything phase C shares
        {
            const bool part = j < CH;                  // lanes taking part in this pass
            const int s = base + j;
            // Everything only phase A needs is re-read here (the linearisation biases from L2, R(q_k_lin) g from
            // LDS) instead of living in registers across phase C: the recursion needs every register it can get.
            long long wq = w;
            asm volatile("" : "+v"(wq));   // opaque to the optimiser: keeps the loads inside the loop
            V3 gk = mk(0, 0, 0);
            if (MODEL == 2) gk = rec_v3(gs, GS_GK);
            SampleRec r;
            if (part && s < n) {
                const double *ka = A.knots + (k0 + s) * 7;
                const V3 bw = ldv3(A.lin + wq * 6), ba = ldv3(A.lin + wq * 6 + 3);
                double a[14];
#pragma unroll
                for (int i = 0; i < 14; i++) a[i] = ka[i];
                r = make_sample_rec<MODEL, AVG>(a[0], a[7], mk(a[1], a[2], a[3]), mk(a[4], a[5], a[6]),
                                                mk(a[8], a[9], a[10]), mk(a[11], a[12], a[13]), bw, ba);
            } else {  // padding: an exact no-op interval
                r.dt = 0; r.w = mk(0, 0, 0); r.a0 = mk(0, 0, 0); r.a1 = mk(0, 0, 0);
                r.f1 = r.f2 = r.f3 = r.f4 = 0; r.Rstep = eye(); r.Rhalf = eye();
            }
            M3 inc = r.Rstep;   // inclusive prefix product (later factors on the left), Hillis-Steele
            M3 pre;
            if constexpr (GROUP == 32) {
                // two DPP rows per group: scan each row with row_shr moves, then fold row 0's total (its lane 15,
                // handed to the next row by row_bcast:15) into row 1 -- no ds_bpermute, no LDS round trips to wait for
                const int jr = j & 15;
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) {
                    const M3 t = row16_up_m3(inc, d);
                    if (jr >= d) inc = mm(inc, t);
                }
                const M3 T0 = row_bcast15_to_odd_rows(inc);
                if (j >= 16) inc = mm(inc, T0);
                pre = row16_up_m3(inc, 1);
                if (j == 16) pre = T0;
            } else {
#pragma unroll
                for (int d = 1; d < CH; d <<= 1) {
                    const M3 t = group_up<GROUP>(inc, d);
                    if (j >= d) inc = mm(inc, t);
                }
                pre = group_up<GROUP>(inc, 1);
            }
            if (j == 0) pre = eye();
            const M3 Rc = rec_mat(gs, GS_R);                       // rotation carried in from the previous chunk
            double *irw = irs + (g * CH + min(j, CH - 1)) * IRD;
            MeanInc mi;
            if (part) mi = finish_interval<MODEL, AVG>(r, mm(pre, Rc), gk, irw);
            else { mi.alpha = mk(0, 0, 0); mi.beta = mk(0, 0, 0); mi.dt = 0; }
            if constexpr (GROUP == 32) {
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) {                 // in-row: lane j <- j (earlier) o j+d (later)
                    MeanInc o;
                    o.beta = group_down<16>(mi.beta, d); o.alpha = group_down<16>(mi.alpha, d);
                    o.dt = group_down<16>(mi.dt, d);
                    if ((j & 15) + d < 16) mi = inc_combine(mi, o);
                }
                MeanInc o;                                         // row 0's total o row 1's total (one LDS shuffle)
                o.beta = shfl_down(mi.beta, 16, GROUP); o.alpha = shfl_down(mi.alpha, 16, GROUP);
                o.dt = __shfl_down(mi.dt, 16, GROUP);
                mi = inc_combine(mi, o);
            } else {
#pragma unroll
                for (int d = 1; d < CH; d <<= 1) {                 // ordered reduction: lane j <- j (earlier) o j+d (later)
                    MeanInc o;
                    o.beta = group_down<GROUP>(mi.beta, d); o.alpha = group_down<GROUP>(mi.alpha, d);
                    o.dt = group_down<GROUP>(mi.dt, d);
                    mi = inc_combine(mi, o);
                }
            }
            wave_lds_fence();   // every lane has read the carried rotation
            if (j == 0) { gs_apply_inc(gs, mi); rec_put_mat(gs, GS_R0, Rc); }
            if (j == CH - 1) rec_put_mat(gs, GS_R, mm(inc, Rc));
        }
        wave_lds_fence();

        // ---- phase C: sequential RK4 recursion over the staged intervals; F x is lane-local, P F^T arrives
        // through the exchange rows
        const int cnt = min(CH, nmax - base);
        M3 Rs = eye();
        for (int sl = 0; sl < cnt; ++sl) {
            const double *ir = irs + (g * CH + sl) * IRD;  // group-uniform address: LDS broadcast
            cov_begin<MODEL>(Ln, ir, hoff);
#pragma unroll
            for (int stg = 0; stg < 4; ++stg) {
                double M[9];
                // The stage rotation is read from the record only when it changes: stages 1 and 2 share R_mid, and the
                // R_new of stage 3 IS the R_old of the next interval's stage 0 (re-read at the start of a pass only).
                // 10 fewer LDS broadcasts per interval: -5 % (model 1), -3 % (model 2).
                if (stg == 0) { if (sl == 0) Rs = rec_mat(gs, GS_R0); }
                else if (stg != 2) Rs = cov_stage_rotation<MODEL>(ir, stg);
                cov_stage_M(Ln, stg, Rs, M);
                if (jj < D::NPCOL) {
#pragma unroll
                    for (int rr = 0; rr < CovExchRows<MODEL>::V; rr++) ex_g[rr * EP + exch_pos<MODEL>(jj)] = M[rr];
                }
                // The exchange is private to this wavefront and a wave's DS instructions execute in issue
                // order, so the row reads below see the writes above without draining lgkmcnt; only the
                // COMPILER must not reorder them (no instruction is emitted here).
                wave_lds_fence();
                if constexpr (CovPBySymmetry<MODEL>::V) {
                    // rows p of F X = rows v of X = (symmetry) the columns the v lanes hold: lanes 12-14 take them from
                    // lanes 6-8 by a masked row_shr:6 instead of through LDS (cpi_math.hpp: CovPBySymmetry)
                    double mt[D::NR];
                    const double *Xs = cov_stage_X(Ln, stg);
#pragma unroll
                    for (int i = 0; i < D::NR; i++) mt[i] = dpp_shr6_bank3(ex_row[exch_pos<MODEL>(i)], Xs[i]);
                    cov_stage_finish_regs(Ln, stg, M, mt);
                } else {
                    cov_stage_finish(Ln, stg, M, ex_row);
                }
            }
            cov_end(Ln);
            if (MODEL == 2) {  // column clone: columns 15:18 := columns 0:3 (CpiV2.h:436-441)
#pragma unroll
                for (int i = 0; i < D::NR; i++) Ln.P0[i] = dpp_clone_shr4(Ln.P0[i]);
            }
        }
        wave_lds_fence();
    }

    if (!valid) return;
    if (A.out.P && jj < 15) {
        double *p = A.out.P + w * 225 + jj * 15;
#pragma unroll
        for (int i = 0; i < 15; i++) p[i] = Ln.P0[i];
    }
    if (A.write_means && j == 0) {
        if (A.out.DT) A.out.DT[w] = gs[GS_DT];
        if (A.out.alpha) stv3(A.out.alpha + w * 3, rec_v3(gs, GS_ALPHA));
        if (A.out.beta) stv3(A.out.beta + w * 3, rec_v3(gs, GS_BETA));
        if (A.out.q) {
            const Q4 q = rot_2_quat(rec_mat(gs, GS_R));
            double *p = A.out.q + w * 4;
            p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w;
        }
    }
    if (MODEL == 2 && A.write_jac && jj >= D::NPCOL && jj < D::NCOL) {
        // Jacobian read-out of Discrete_J_b (CpiV2.h:450-458); d = which column, c = column within the block
        const int d = (jj - D::NPCOL) / 3, c = (jj - D::NPCOL) % 3;
        const V3 th = mk(Ln.P0[0], Ln.P0[1], Ln.P0[2]);
        const V3 vv = mk(Ln.P0[6], Ln.P0[7], Ln.P0[8]);
        const V3 pp = mk(Ln.P0[12], Ln.P0[13], Ln.P0[14]);
        if (d == 0) {
            if (A.out.J_q) stv3(A.out.J_q + w * 9 + c * 3, -th);
            if (A.out.J_a) stv3(A.out.J_a + w * 9 + c * 3, pp);
            if (A.out.J_b) stv3(A.out.J_b + w * 9 + c * 3, vv);
        } else if (d == 1) {
            if (A.out.H_a) stv3(A.out.H_a + w * 9 + c * 3, pp);
            if (A.out.H_b) stv3(A.out.H_b + w * 9 + c * 3, vv);
        } else {
            if (A.out.O_a) stv3(A.out.O_a + w * 9 + c * 3, pp);
            if (A.out.O_b) stv3(A.out.O_b + w * 9 + c * 3, vv);
        }
    }
}

// ============================================================================================
// Forster / GTSAM discrete-preintegration comparator kernel (fsd:: in cpi_math.hpp)
// ============================================================================================
// Replaces: the PreintegratedCombinedMeasurements loop of GraphSolver::createimufactor_discrete
// (GraphSolver_IMU.cpp:149-199) and its call-site conversions (:204-225, swapcovariance :240-254).
// 16 lanes per window, 4 windows per wavefront.  Lane j < 15 owns column j of the 15x15 covariance (already in the
// block order [theta b_g v b_a p] the call site swaps it into); lanes 0-2 also carry column j of the three gyro-bias
// Jacobians, lanes 3-5 column j-3 of the two accelerometer-bias Jacobians; every lane carries the means (the SIMD
// cost is the same as one lane doing it).  P' = F P F^T + G per interval: F x is lane-local (F is sparse), the
// transposed product arrives through ONE 9-row LDS exchange per interval (the continuous models need four, one per
// RK4 stage).  F depends on the interval alone, so phase A (one lane per interval: Exp, its right Jacobian) needs no
// prefix scan.
__global__ __launch_bounds__(64, 2) void cpi_forster_kernel(PreArgs A) {
    // 12 intervals per phase-A pass: 9.8 KB of records + 9 KB of exchange rows per wavefront, two wavefronts per SIMD.
    // LDS banking (64 x 4 B; ds_read_b128 serves mixed 16-lane groups of two windows, MI355X_MICROARCH.md "LDS"):
    //  * records are pitched 26 doubles (208 B), so the 8 lanes of a ds_write_b128 group land on distinct 16-B slots
    //    in phase A and the four windows' broadcast reads of "their" record (window stride 12 x 208 B = 192 mod 256)
    //    use different slots -- with the natural 24-double pitch all four windows hit the same banks (2-way conflict
    //    on every record read);
    //  * exchange rows are pitched 18 doubles (144 B = 9 slots) and windows 288 doubles (0 mod 256 B): the two
    //    half-windows a ds_read_b128 lane group mixes then read complementary slot sets.
    constexpr int GROUP = 16, G = 64 / GROUP, CH = 12, EP = EXCH_PITCH, IRD = 26, ROWS = 15, EXW = 16 * EP;
    static_assert(IRD >= fsd::IR_SIZE, "record pitch");
    __shared__ __attribute__((aligned(256))) double irs[G * CH * IRD];   // interval records
    __shared__ __attribute__((aligned(256))) double exch[G * EXW];       // row exchange: (F P) of each window

    const int lane = threadIdx.x;
    const int g = lane / GROUP, j = lane % GROUP;
    long long w = (long long)blockIdx.x * G + g;
    const bool valid = w < A.W;
    if (!valid) w = A.W - 1;
    const int n = A.count ? min(max(A.count[w], 0), A.N) : A.N;
    const long long k0 = A.first ? A.first[w] : w * (long long)(A.N + 1);
    const int nmax = wave_max(n);

    const double q4[4] = { A.q4[0], A.q4[1], A.q4[2], A.q4[3] };
    // per-lane constant vectors instead of selects inside the recursion
    const V3 eg = (j < 3) ? unit(j) : mk(0, 0, 0);                 // gyro-bias Jacobian column j
    const V3 ek = (j >= 3 && j < 6) ? unit(j - 3) : mk(0, 0, 0);   // accelerometer-bias Jacobian column j - 3
    const double th_on = (j < 3) ? 1.0 : 0.0;
    // process noise on this column's own diagonal entry, as one vector per constant-diagonal block
    const V3 nbg = (j >= 3 && j < 6) ? q4[1] * unit(j - 3) : mk(0, 0, 0);
    const V3 nv = (j >= 6 && j < 9) ? q4[2] * unit(j - 6) : mk(0, 0, 0);
    const V3 nba = (j >= 9 && j < 12) ? q4[3] * unit(j - 9) : mk(0, 0, 0);
    double *ex_g = exch + g * EXW;
    // Column j of P F^T is ROW j of F P.  The bias rows of F are identity rows, so for a bias column that row is the
    // lane's own column -- it is still written and read back like the others: 6 more LDS writes per lane cost less
    // than 30 v_cndmask per interval on the VALU, which is what bounds this kernel.  (Lane 15 owns nothing: it runs
    // as a shadow of column 0 and never writes.)
    const double *ex_row = ex_g + (j < 15 ? j : 0) * EP;
    const int jdrow = fsd::IR_JD + 3 * min(j, 2);

    fsd::Mean m;
    fsd::JacCol J;
    fsd::mean_init(m);
    fsd::jac_init(J);
    double x[15];
#pragma unroll
    for (int i = 0; i < 15; i++) x[i] = 0.0;

    // The knots of the NEXT phase-A pass are requested before phase C of the current one and only consumed after
    // it: with two wavefronts per SIMD an exposed HBM round trip per 12 intervals was a quarter of the kernel's time
    // (0.78 -> see DESIGN.md).  vmcnt and lgkmcnt are separate counters, so phase C's LDS waits do not drain them.
    const V3 bgl = ldv3(A.lin + w * 6), bal = ldv3(A.lin + w * 6 + 3);
    double kn[8];
#pragma unroll
    for (int i = 0; i < 8; i++) kn[i] = 0.0;
    if (j < CH && j < n) {
        const double *ka = A.knots + (k0 + j) * 7;
#pragma unroll
        for (int i = 0; i < 8; i++) kn[i] = ka[i];
    }
    for (int base = 0; base < nmax; base += CH) {
        if (j < CH) {   // ---- phase A: lane (g, j) builds the record of interval base + j of its window
            const int s = base + j;
            fsd::Rec r;
            if (s < n) {
                r = fsd::make_rec(kn[0], kn[7], mk(kn[1], kn[2], kn[3]), mk(kn[4], kn[5], kn[6]), bgl, bal, q4[0]);
            } else {
                r.dt = 0; r.qs = 0; r.a = mk(0, 0, 0); r.E = eye(); r.JD = zero3();
            }
            fsd::put_rec(irs + (g * CH + j) * IRD, r);
            if (s + CH < n) {
                const double *ka = A.knots + (k0 + s + CH) * 7;
#pragma unroll
                for (int i = 0; i < 8; i++) kn[i] = ka[i];
            }
        }
        wave_lds_fence();

        // ---- phase C: the sequential recursion over the staged intervals
        const int cnt = min(CH, nmax - base);
        for (int sl = 0; sl < cnt; ++sl) {
            const double *ir = irs + (g * CH + sl) * IRD;   // group-uniform address: LDS broadcast
            const fsd::Rec r = fsd::get_rec(ir);
            fsd::jac_step(J, m.R, r, ek, eg);               // uses the rotation BEFORE this interval
            fsd::mean_step(m, r);
            double y[15];
            fsd::F_apply(r, x, y);
            if (j < 15) {
#pragma unroll
                for (int i = 0; i < 15; i++) ex_g[i * EP + j] = y[i];
            }
            wave_lds_fence();   // DS instructions of a wave execute in order; this only pins the compiler
            double z[15];
#pragma unroll
            for (int i = 0; i < 15; i++) z[i] = ex_row[i];
            fsd::F_apply(r, z, x);
            fsd::theta_noise_add(x, r, rec_v3(ir, jdrow), th_on);
            x[3] = fma(r.dt, nbg.x, x[3]); x[4] = fma(r.dt, nbg.y, x[4]); x[5] = fma(r.dt, nbg.z, x[5]);
            x[6] = fma(r.dt, nv.x, x[6]); x[7] = fma(r.dt, nv.y, x[7]); x[8] = fma(r.dt, nv.z, x[8]);
            x[9] = fma(r.dt, nba.x, x[9]); x[10] = fma(r.dt, nba.y, x[10]); x[11] = fma(r.dt, nba.z, x[11]);
            wave_lds_fence();
        }
    }

    const int kind = (j < 3) ? 0 : ((j < 6) ? 1 : 2), jc = (j < 3) ? j : ((j < 6) ? j - 3 : 0);
    if (!valid) return;
    if (A.out.P && j < 15) {
        double *p = A.out.P + w * 225 + j * 15;
#pragma unroll
        for (int i = 0; i < 15; i++) p[i] = x[i];
    }
    if (j == 0) {
        if (A.out.DT) A.out.DT[w] = m.dT;
        if (A.out.alpha) stv3(A.out.alpha + w * 3, m.p);      // deltaPij (:204)
        if (A.out.beta) stv3(A.out.beta + w * 3, m.v);        // deltaVij (:205)
        if (A.out.q) {                                        // rot_2_quat(deltaRij^T) (:206, :229)
            M3 Rt;
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int k = 0; k < 3; k++) Rt.m[i][k] = m.R.m[k][i];
            const Q4 q = rot_2_quat(Rt);
            double *p = A.out.q + w * 4;
            p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w;
        }
    }
    if (kind == 0) {
        if (A.out.J_q) stv3(A.out.J_q + w * 9 + jc * 3, -J.r);   // -delRdelBiasOmega (:210)
        if (A.out.J_a) stv3(A.out.J_a + w * 9 + jc * 3, J.p);    // delPdelBiasOmega (:212)
        if (A.out.J_b) stv3(A.out.J_b + w * 9 + jc * 3, J.v);    // delVdelBiasOmega (:214)
    } else if (kind == 1) {
        if (A.out.H_a) stv3(A.out.H_a + w * 9 + jc * 3, J.p);    // delPdelBiasAcc (:211)
        if (A.out.H_b) stv3(A.out.H_b + w * 9 + jc * 3, J.v);    // delVdelBiasAcc (:213)
    }
}

// ============================================================================================
// factor kernels
// ============================================================================================
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
};

__device__ __forceinline__ NavState ld_state(const double *p) {
    NavState s;
    s.q = ldq4(p); s.bg = ldv3(p + 4); s.v = ldv3(p + 7); s.ba = ldv3(p + 10); s.p = ldv3(p + 13);
    return s;
}

// One SoA input field (K doubles per factor) of the FPW consecutive factors of a wavefront: FPW*K contiguous
// doubles, lane i takes doubles i, i + 64, ...  load() is unconditional (index clamped to the last valid double),
// store() writes record-major into the LDS staging area.
template <int FPW, int K>
struct FieldFetch {
    static constexpr int R = (FPW * K + 63) / 64;
    double v[R];
    __device__ __forceinline__ void load(const double *src, long long f0, int nf, int lane) {
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = src[f0 * K + min(lane + 64 * r, nf * K - 1)];
    }
    __device__ __forceinline__ void store(double *sIn, int pitch, int off, int lane) const {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = lane + 64 * r, g = i / K;
            if (i < FPW * K) sIn[g * pitch + off + (i - g * K)] = v[r];
        }
    }
};

// Record layout of one factor in the LDS staging area (doubles)
namespace fin {
constexpr int O_ALPHA = 0, O_BETA = 3, O_Q = 6, O_LIN = 10, O_JQ = 16, O_JB = 25, O_JA = 34, O_HB = 43, O_HA = 52,
              O_DT = 61, O_QK = 62, O_OB = 66, O_OA = 75, O_XI = 84, O_XJ = 100, IN_D = 116;
}
template <int MODEL, int FPW, bool WHITEN>
__device__ __forceinline__ void factor_fetch_inputs(const FactorArgs &A, long long f0, int nf, int lane, double *sIn,
                                                    double *sR) {
    using namespace fin;
    constexpr bool whiten = WHITEN;
    constexpr int HB = FPW * 225;
    // ---- cooperative, de-duplicated input fetch: every double of the FPW factors' records is loaded from HBM
    // exactly once per wavefront (consecutive lanes = consecutive doubles of one SoA field) into LDS, from
    // where the lanes of a factor read it as broadcasts.  All loads are issued unconditionally (clamped
    // addresses) before the first LDS write, so the wavefront pays ONE memory latency (two for the states
    // when they are gathered through idx_i / idx_j), not one per field.
    {
        constexpr int SR = (FPW * 16 + 63) / 64;
        long long si[SR], sj[SR];
#pragma unroll
        for (int r = 0; r < SR; r++) {
            const int i = min(lane + 64 * r, FPW * 16 - 1), g = i >> 4;
            const long long ff = min(f0 + g, A.F - 1);
            // branch-free NULL handling (a valid dummy address is read and discarded) keeps all loads in one block
            const int vi = (A.idx_i ? A.idx_i : reinterpret_cast<const int *>(A.states))[ff];
            const int vj = (A.idx_j ? A.idx_j : reinterpret_cast<const int *>(A.states))[ff];
            si[r] = min(max(A.idx_i ? (long long)vi : ff, 0ll), A.S - 1);
            sj[r] = min(max(A.idx_j ? (long long)vj : ff + 1, 0ll), A.S - 1);
        }
        FieldFetch<FPW, 3> f_alpha, f_beta; FieldFetch<FPW, 4> f_q, f_qk; FieldFetch<FPW, 6> f_lin;
        FieldFetch<FPW, 9> f_jq, f_jb, f_ja, f_hb, f_ha, f_ob, f_oa; FieldFetch<FPW, 1> f_dt;
        f_alpha.load(A.meas.alpha, f0, nf, lane); f_beta.load(A.meas.beta, f0, nf, lane); f_q.load(A.meas.q, f0, nf, lane);
        f_lin.load(A.lin, f0, nf, lane); f_jq.load(A.meas.J_q, f0, nf, lane); f_jb.load(A.meas.J_b, f0, nf, lane);
        f_ja.load(A.meas.J_a, f0, nf, lane); f_hb.load(A.meas.H_b, f0, nf, lane); f_ha.load(A.meas.H_a, f0, nf, lane);
        f_dt.load(A.meas.DT, f0, nf, lane);
        if (MODEL == 2) { f_qk.load(A.qk, f0, nf, lane); f_ob.load(A.meas.O_b, f0, nf, lane); f_oa.load(A.meas.O_a, f0, nf, lane); }
        double xi[SR], xj[SR];
#pragma unroll
        for (int r = 0; r < SR; r++) {
            const int e = lane & 15;
            xi[r] = A.states[si[r] * 16 + e];
            xj[r] = A.states[sj[r] * 16 + e];
        }
        double R_[WHITEN ? (HB + 63) / 64 : 1];
        if (whiten) {
#pragma unroll
            for (int r = 0; r < (HB + 63) / 64; r++) R_[r] = A.sqrt_info[f0 * 225 + min(lane + 64 * r, nf * 225 - 1)];
        }
        f_alpha.store(sIn, IN_D, O_ALPHA, lane); f_beta.store(sIn, IN_D, O_BETA, lane); f_q.store(sIn, IN_D, O_Q, lane);
        f_lin.store(sIn, IN_D, O_LIN, lane); f_jq.store(sIn, IN_D, O_JQ, lane); f_jb.store(sIn, IN_D, O_JB, lane);
        f_ja.store(sIn, IN_D, O_JA, lane); f_hb.store(sIn, IN_D, O_HB, lane); f_ha.store(sIn, IN_D, O_HA, lane);
        f_dt.store(sIn, IN_D, O_DT, lane);
        if (MODEL == 2) { f_qk.store(sIn, IN_D, O_QK, lane); f_ob.store(sIn, IN_D, O_OB, lane); f_oa.store(sIn, IN_D, O_OA, lane); }
#pragma unroll
        for (int r = 0; r < SR; r++) {
            const int i = lane + 64 * r, g = i >> 4, e = i & 15;
            if (i < FPW * 16) { sIn[g * IN_D + O_XI + e] = xi[r]; sIn[g * IN_D + O_XJ + e] = xj[r]; }
        }
        if (whiten) {
#pragma unroll
            for (int r = 0; r < (HB + 63) / 64; r++)
                if (lane + 64 * r < HB) sR[lane + 64 * r] = R_[r];
        }
    }
}
// All fields of a staged record, by reference (read where they are used).
__device__ __forceinline__ FactorMeas factor_meas_of(const double *in, const double grav[3]) {
    using namespace fin;
    FactorMeas m;
    m.alpha = in + O_ALPHA; m.beta = in + O_BETA; m.q_KtoK1 = in + O_Q; m.lin = in + O_LIN; m.J_q = in + O_JQ;
    m.J_beta = in + O_JB; m.J_alpha = in + O_JA; m.H_beta = in + O_HB; m.H_alpha = in + O_HA; m.dt = in + O_DT;
    m.q_K_lin = in + O_QK; m.O_beta = in + O_OB; m.O_alpha = in + O_OA; m.xi = in + O_XI; m.xj = in + O_XJ;
    m.grav = mk(grav[0], grav[1], grav[2]);
    return m;
}

// LPF lanes per factor (16, 8 or 4), FPW = 64 / LPF factors per wavefront.  Every lane evaluates the shared
// quaternion algebra of its factor (so it is done LPF times per factor); lane q of a factor then owns columns
// q, q + LPF, ... of H1 / H2.  The sweep is HBM-WRITE bound (3 720 of 4 496 B per factor are the dense 15x15
// pair), so the columns are transposed through LDS and leave the wavefront as full, consecutive 16-byte stores:
// the FPW factors' H1 blocks are one contiguous FPW x 1 800-byte span of the output (measured on MI355X: that
// pattern stores at 4.8 TB/s, per-column 120/240-byte pieces at 2.5 TB/s -- which rules out one lane per factor).
// LPF = 16 has the most wavefronts (small sweeps fill the chip); LPF = 8 / 4 do 2x / 4x less redundant arithmetic.
#ifndef CPI_FACTOR_WPS
#define CPI_FACTOR_WPS 1
#endif
template <int MODEL, bool WHITEN, int LPF>
__global__ __launch_bounds__(64, CPI_FACTOR_WPS) void cpi_factor_kernel(FactorArgs A) {
    constexpr int FPW = 64 / LPF;                // factors per wavefront
    constexpr int CPL = (15 + LPF - 1) / LPF;    // columns per lane
    constexpr int HB = FPW * 225;                // doubles of H1 (or H2) per wavefront
    constexpr int IN_D = fin::IN_D;
    __shared__ __attribute__((aligned(16))) double sH[HB + FPW * 15 + 4];   // one 15x15 set at a time: H1, then H2
    __shared__ __attribute__((aligned(16))) double sIn[FPW * IN_D];         // the factors' input records
    __shared__ __attribute__((aligned(16))) double sR[WHITEN ? HB : 2];     // whitening only: the factors' R
    const int lane = threadIdx.x;
    const int q = lane % LPF, fl = lane / LPF;
    const long long f0 = (long long)blockIdx.x * FPW;
    const int nf = (int)min((long long)FPW, A.F - f0);
    constexpr bool whiten = WHITEN;

    factor_fetch_inputs<MODEL, FPW, WHITEN>(A, f0, nf, lane, sIn, sR);
    __syncthreads();
    const double *in = sIn + fl * IN_D;
    const FactorMeas m = factor_meas_of(in, A.grav);   // every field is read from LDS where it is used
    double *s1 = sH, *se = sH + HB;
    const double *Rf = sR + fl * 225;
    const int nd = nf * 225, ne = nf * 15;

    // ---- shared algebra; the residual goes to the staging area at once.  Lane q of a factor publishes rows
    // q, q + LPF, ... of the 15-vector.
    FactorShared S;
    {
        V3 e5[5];
        factor_shared_core<MODEL>(m, S, e5);
#pragma unroll
        for (int k = 0; k < CPL; k++) {
            const int c = q + LPF * k;
            if (c < 15) {
                const V3 ec = pick5(e5[0], e5[1], e5[2], e5[3], e5[4], c / 3);
                se[fl * 15 + c] = sel3(ec.x, ec.y, ec.z, c % 3);
            }
        }
    }
    // ---- optional whitening (GTSAM Gaussian::WhitenSystem): y = R x with R upper triangular, column-major
    auto whiten_col = [&](double *h) {   // in place: out[i] = sum_{k >= i} R[i][k] h[k]
#pragma unroll
        for (int i = 0; i < 15; i++) {
            double acc = 0.0;
#pragma unroll
            for (int k = i; k < 15; k++) acc = fma(Rf[k * 15 + i], h[k], acc);
            h[i] = acc;                   // rows are finished top-down, so h[k], k > i, is still unwhitened
        }
    };
    if (whiten) {   // R err needs the whole residual
        wave_lds_fence();
        double acc[CPL];
#pragma unroll
        for (int k = 0; k < CPL; k++) {
            const int cr = min(q + LPF * k, 14);
            acc[k] = 0.0;
            for (int j = 0; j < 15; j++) acc[k] = fma((j >= cr) ? Rf[j * 15 + cr] : 0.0, se[fl * 15 + j], acc[k]);
        }
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < CPL; k++) {
            const int c = q + LPF * k;
            if (c < 15) se[fl * 15 + c] = acc[k];
        }
    }

    // ---- this lane's columns -> LDS (layout identical to the global layout of this wavefront's span) -> HBM,
    // consecutive lanes = consecutive 16-byte pieces (gfx950 global memory takes dwordx4 at 8-byte alignment).
    // H1 and H2 take turns in the same staging area to keep LDS per wavefront small.
    struct __attribute__((packed, aligned(8))) d2u { double a, b; };
    auto flush = [&](double *dst, const double *src, int n) {
        const int n2 = n >> 1;
        for (int i = lane; i < n2; i += 64) { d2u v; v.a = src[2 * i]; v.b = src[2 * i + 1]; ((d2u *)dst)[i] = v; }
        if ((n & 1) && lane == 0) dst[n - 1] = src[n - 1];
    };
    const Q4 qi = ldq(m.xi);
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        double *H = pass == 0 ? A.H1 : A.H2;
        if (H) {
            if (pass == 1 && A.H1) wave_lds_fence();   // in-order DS: the H1 flush reads complete before these writes land
#pragma unroll
            for (int k = 0; k < CPL; k++) {
                // (Measured: giving lane q the ADJACENT columns CPL*q + k instead removes the 2-way LDS bank conflict of
                // these column writes -- a 16-lane ds_write_b64 group holds two factors one 8-byte slot apart --
                // SQ_LDS_BANK_CONFLICT -83 %, but the sweep is HBM-write bound and gets no faster: A/B on one box
                // 0.88 / 0.89-0.92 ms vs 0.85-0.89 / 0.90-0.91 ms per 1 M factors (model 1 / 2).  Kept interleaved.)
                const int c = q + LPF * k;
                if (c < 15) {
                    double h[15];
                    S.bc = c / 3; S.cc = c - 3 * S.bc;
                    S.u = unit(S.cc);
                    S.rku = qrot(qi, S.u);      // column cc of quat_2_Rot(q_GtoK)
                    if (pass == 0) factor_H1_column<MODEL>(S, m, h);
                    else factor_H2_column(S, h);
                    if (whiten) whiten_col(h);
#pragma unroll
                    for (int i = 0; i < 15; i++) s1[fl * 225 + c * 15 + i] = h[i];
                }
            }
        }
        if (pass == 0) {
            wave_lds_fence();
            flush(A.err + f0 * 15, se, ne);
            if (A.H1) flush(A.H1 + f0 * 225, s1, nd);
        } else if (H) {
            wave_lds_fence();
            flush(A.H2 + f0 * 225, s1, nd);
        }
    }
}

// Packed evaluateError: only what depends on the current states (cpi_factor_eval_packed_batch, include/cpi_amd.h).
// Of the 450 doubles of the dense H1 / H2 pair, 54 depend on the states -- the 3x3 blocks H1(0,0), H1(6,0),
// H1(12,0), H1(0,3), H2(0,0) and R(q_GtoK), which appears five times; the rest is 0, +-I or a copy of a measurement
// field the caller already holds.  72 doubles per factor (15 residual + 6 blocks + 3 of padding, 576 B = 36 x 16 B)
// instead of 465: the sweep stops being bound by the write of mostly-constant matrices.
// LPF lanes per factor: lane q owns the columns q, q + LPF, ... < 6 of H1 (column c < 3: blocks (0,0), (6,0), (12,0), plus
// column c of H2(0,0) and of R(q_GtoK); 3 <= c < 6: block (0,3)) and the residual rows q, q + LPF, ... < 15.
constexpr int FACTOR_PACKED_DOUBLES = 72;
template <int MODEL, int LPF>
__global__ __launch_bounds__(64) void cpi_factor_packed_kernel(FactorArgs A, double *packed) {
    constexpr int FPW = 64 / LPF, PD = FACTOR_PACKED_DOUBLES, IN_D = fin::IN_D;
    __shared__ __attribute__((aligned(16))) double sP[FPW * PD];
    __shared__ __attribute__((aligned(16))) double sIn[FPW * IN_D];
    __shared__ double sDummy[2];
    const int lane = threadIdx.x;
    const int q = lane % LPF, fl = min(lane / LPF, FPW - 1);   // 64 mod LPF spare lanes repeat the last factor's lane 0 (same values, same slots)
    const long long f0 = (long long)blockIdx.x * FPW;
    const int nf = (int)min((long long)FPW, A.F - f0);
    factor_fetch_inputs<MODEL, FPW, false>(A, f0, nf, lane, sIn, sDummy);
    __syncthreads();
    const double *in = sIn + fl * IN_D;
    const FactorMeas m = factor_meas_of(in, A.grav);
    double *out = sP + fl * PD;
    FactorShared S;
    {
        V3 e5[5];
        factor_shared_core<MODEL>(m, S, e5);
#pragma unroll
        for (int k = 0; k < (15 + LPF - 1) / LPF; k++) {
            const int c = q + LPF * k;
            if (c < 15) {
                const V3 ec = pick5(e5[0], e5[1], e5[2], e5[3], e5[4], c / 3);
                out[c] = sel3(ec.x, ec.y, ec.z, c % 3);
            }
        }
    }
    const Q4 qi = ldq(m.xi);
#pragma unroll
    for (int k = 0; k < (6 + LPF - 1) / LPF; k++) {
        const int c = q + LPF * k;               // column c of H1: 0..2 -> blocks (0,0), (6,0), (12,0) (+ H2(0,0), R(q_GtoK)); 3..5 -> block (0,3)
        if (c < 6) {
            double h[15];
            S.bc = c / 3; S.cc = c - 3 * S.bc;
            S.u = unit(S.cc);
            S.rku = qrot(qi, S.u);
            factor_H1_column<MODEL>(S, m, h);
            if (c < 3) {
                double *b = out + 15 + 3 * c;                    // H1(0,0), H1(6,0), H1(12,0): column c of each
                b[0] = h[0]; b[1] = h[1]; b[2] = h[2];
                b[9] = h[6]; b[10] = h[7]; b[11] = h[8];
                b[18] = h[12]; b[19] = h[13]; b[20] = h[14];
                double h2[15];
                factor_H2_column(S, h2);
                double *r = out + 51 + 3 * c;                    // R(q_GtoK) column c, then H2(0,0) column c
                r[0] = S.rku.x; r[1] = S.rku.y; r[2] = S.rku.z;
                r[9] = h2[0]; r[10] = h2[1]; r[11] = h2[2];
            } else {
                double *b = out + 42 + 3 * (c - 3);              // H1(0,3) column c - 3
                b[0] = h[0]; b[1] = h[1]; b[2] = h[2];
            }
        }
    }
    if (q == LPF - 1) { out[69] = 0.0; out[70] = 0.0; out[71] = 0.0; }
    wave_lds_fence();
    struct __attribute__((packed, aligned(8))) d2u { double a, b; };
    d2u *dst = reinterpret_cast<d2u *>(packed + f0 * PD);
    for (int i = lane; i < nf * (PD / 2); i += 64) { d2u v; v.a = sP[2 * i]; v.b = sP[2 * i + 1]; dst[i] = v; }
}

// R = chol_upper(P^-1) = B^-1 with P = B B^T, B upper triangular ("reverse" Cholesky, from the last pivot up).
// 16 lanes (one DPP row) per factor, 4 factors per wavefront; lane j keeps the FULL symmetric column j of the
// working matrix in registers, so its own B[j][k] is a static register (a[k]) and the only cross-lane traffic is
// "every lane reads column k of lane k": DPP row_share broadcasts, no LDS in the factorisation.  The inverse of
// the triangular factor is fused into the same sweep: back substitution for column j of U = B^-1,
//   U[j][j] = 1/B[j][j],   U[k][j] = -(sum_{m=k+1..j} B[k][m] U[m][j]) / B[k][k]   (k < j),   0 below the diagonal,
// consumes the columns of B in the order the factorisation produces them (k = 14 .. 0), so every lane folds the
// broadcast column k into its running sums acc[i] = sum_m B[i][m] U[m][j] right away and B is never stored.
// Input and output pass through LDS so that HBM sees full consecutive 16-byte pieces (see cpi_factor_kernel).
template <int K>
__device__ __forceinline__ double row_share(double v) {   // all 16 lanes of a DPP row read lane K of that row
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int nlo = __builtin_amdgcn_mov_dpp(lo, 0x150 + K, 0xf, 0xf, false);
    const int nhi = __builtin_amdgcn_mov_dpp(hi, 0x150 + K, 0xf, 0xf, false);
    return __hiloint2double(nhi, nlo);
}
template <int K>
__device__ __forceinline__ void chol_inv_step(double (&a)[15], double (&u)[15], double (&acc)[15], int j) {
    if constexpr (K >= 0) {
        // pivot: b_kk = sqrt(A[k][k]); a non-positive (or NaN) pivot poisons the factor with NaNs
        const double akk = row_share<K>(a[K]);
        double bkk, inv;
        mag_and_inverse(akk, bkk, inv);
        if (!(akk > 0.0)) inv = __builtin_nan("");
        // row k of column j of U
        u[K] = (K == j) ? inv : ((K < j) ? -acc[K] * inv : 0.0);
        // lanes j < k: trailing update of column j (all rows < k) with B[j][k] = A[k][j] / b_kk (symmetry: a
        // static register of lane j); finished lanes multiply by zero
        const double bjk = (j < K) ? a[K] * inv : 0.0;
        const double uk = u[K];
#pragma unroll
        for (int i = 0; i < K; i++) {
            const double c = row_share<K>(a[i]) * inv;     // B[i][k], i < k
            a[i] = fma(-c, bjk, a[i]);
            acc[i] = fma(c, uk, acc[i]);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the steps in order: hoisted broadcasts would cost ~200 registers
        chol_inv_step<K - 1>(a, u, acc, j);
    }
}
__global__ __launch_bounds__(64, 3) void cpi_sqrt_info_kernel(long long F, const double *P, double *Rout) {
    constexpr int FPW = 4;
    __shared__ __attribute__((aligned(16))) double sA[FPW * 225];
    const int lane = threadIdx.x, j = lane & 15, fl = lane >> 4;
    const long long f0 = (long long)blockIdx.x * FPW;
    const int nf = (int)min((long long)FPW, F - f0);
    struct __attribute__((packed, aligned(8))) d2u { double a, b; };
    {
        const int n2 = (nf * 225) >> 1;
        const d2u *src = reinterpret_cast<const d2u *>(P + f0 * 225);
        for (int i = lane; i < n2; i += 64) { const d2u v = src[i]; sA[2 * i] = v.a; sA[2 * i + 1] = v.b; }
        if (((nf * 225) & 1) && lane == 0) sA[nf * 225 - 1] = P[f0 * 225 + nf * 225 - 1];
    }
    wave_lds_fence();
    const int fc = min(fl, nf - 1), jc = min(j, 14);   // idle lanes (j == 15, missing factors) redo a valid column
    double a[15], u[15], acc[15];
#pragma unroll
    for (int i = 0; i < 15; i++) { a[i] = sA[fc * 225 + jc * 15 + i]; acc[i] = 0.0; }
    chol_inv_step<14>(a, u, acc, j);
    wave_lds_fence();   // the column reads above are complete (in-order DS) before the staging area is reused
    if (j < 15 && fl < nf) {
#pragma unroll
        for (int i = 0; i < 15; i++) sA[fl * 225 + j * 15 + i] = u[i];
    }
    wave_lds_fence();
    {
        const int n2 = (nf * 225) >> 1;
        d2u *dst = reinterpret_cast<d2u *>(Rout + f0 * 225);
        for (int i = lane; i < n2; i += 64) { d2u v; v.a = sA[2 * i]; v.b = sA[2 * i + 1]; dst[i] = v; }
        if (((nf * 225) & 1) && lane == 0) Rout[f0 * 225 + nf * 225 - 1] = sA[nf * 225 - 1];
    }
}

// Hessian contribution of a factor: what GTSAM's linear solver consumes after
// NoiseModelFactor::linearize (ImuFactorCPIv1.h:82 -> Gaussian::WhitenSystem -> JacobianFactor [A1 A2 | b] with
// A1 = R H1, A2 = R H2, b = -R e) when it builds a HessianFactor: the augmented information matrix
//     [A1 A2 b]^T [A1 A2 b]  =  [ G  g ]      G = A^T A (30x30),  g = A^T b,  f = b^T b
//                               [ g^T f ]
// 31x31 symmetric, written as its packed upper triangle (column-major packed, LAPACK 'U': entry (i, d), i <= d, at
// i + d (d + 1) / 2), 496 doubles per factor.  Fused into the whitened sweep: the 31 whitened columns never leave the
// chip.  16 lanes per factor: lane q < 15 produces columns q of A1 and of A2 (as cpi_factor_kernel), lane 15 the b column;
// lane q then owns columns q and 30 - q of the result (lane 15: column 15): every lane forms 32 dot products' worth of
// useful output from two columns held in registers against the 31 columns broadcast from LDS.
constexpr int HESS_PACKED = 496;
template <int MODEL>
__global__ __launch_bounds__(64, 1) void cpi_factor_hessian_kernel(FactorArgs A, double *hess) {
    constexpr int LPF = 16, FPW = 4, IN_D = fin::IN_D, CP = 16;   // CP: LDS pitch of a whitened column (15 used, 16-B aligned)
    // the packed output stage re-uses the input records and the R matrices: both are dead once the whitened columns sit
    // in sA (one wavefront per workgroup: program order + the LDS fence below order the re-use) -- 32 KB instead of 43
    constexpr int IO_D = (FPW * IN_D + FPW * 225 > FPW * HESS_PACKED) ? FPW * IN_D + FPW * 225 : FPW * HESS_PACKED;
    __shared__ __attribute__((aligned(16))) double sIO[IO_D];
    __shared__ __attribute__((aligned(16))) double sA[FPW * 31 * CP];   // [factor][column][row]
    double *sIn = sIO, *sR = sIO + FPW * IN_D, *sP = sIO;
    const int lane = threadIdx.x;
    const int q = lane % LPF, fl = lane / LPF;
    const long long f0 = (long long)blockIdx.x * FPW;
    const int nf = (int)min((long long)FPW, A.F - f0);
    factor_fetch_inputs<MODEL, FPW, true>(A, f0, nf, lane, sIn, sR);
    __syncthreads();
    const double *in = sIn + fl * IN_D;
    const FactorMeas m = factor_meas_of(in, A.grav);
    const double *Rf = sR + fl * 225;
    double *Af = sA + fl * 31 * CP;
    auto whiten_col = [&](double *h) {   // in place: out[i] = sum_{k >= i} R[i][k] h[k]  (R upper triangular, column-major)
#pragma unroll
        for (int i = 0; i < 15; i++) {
            double acc = 0.0;
#pragma unroll
            for (int k = i; k < 15; k++) acc = fma(Rf[k * 15 + i], h[k], acc);
            h[i] = acc;
        }
    };
    FactorShared S;
    {
        V3 e5[5];
        factor_shared_core<MODEL>(m, S, e5);
        if (q == 15) {   // b = -R e
            double h[15];
#pragma unroll
            for (int b = 0; b < 5; b++) { h[3 * b] = -e5[b].x; h[3 * b + 1] = -e5[b].y; h[3 * b + 2] = -e5[b].z; }
            whiten_col(h);
#pragma unroll
            for (int i = 0; i < 15; i++) Af[30 * CP + i] = h[i];
        }
    }
    if (q < 15) {
        const Q4 qi = ldq(m.xi);
        double h[15];
        S.bc = q / 3; S.cc = q - 3 * S.bc;
        S.u = unit(S.cc);
        S.rku = qrot(qi, S.u);
        factor_H1_column<MODEL>(S, m, h);
        whiten_col(h);
#pragma unroll
        for (int i = 0; i < 15; i++) Af[q * CP + i] = h[i];
        factor_H2_column(S, h);
        whiten_col(h);
#pragma unroll
        for (int i = 0; i < 15; i++) Af[(15 + q) * CP + i] = h[i];
    }
    wave_lds_fence();
    // ---- lane q: columns d1 = q and d2 = 30 - q (lane 15: d1 = d2 = 15) against every column c, broadcast from LDS
    const int d1 = q, d2 = 30 - q;
    double c1[15], c2[15];
#pragma unroll
    for (int i = 0; i < 15; i++) { c1[i] = Af[d1 * CP + i]; c2[i] = Af[d2 * CP + i]; }
    double *Pf = sP + fl * HESS_PACKED;
    const int o1 = d1 * (d1 + 1) / 2, o2 = d2 * (d2 + 1) / 2;
#pragma unroll
    for (int c = 0; c < 31; c++) {
        double a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int i = 0; i < 15; i++) {
            const double x = Af[c * CP + i];   // same address for the 16 lanes of a factor: LDS broadcast
            if (c <= 15) a1 = fma(x, c1[i], a1);   // column d1 <= 15 ends at its diagonal: rows 16..30 belong to other lanes
            a2 = fma(x, c2[i], a2);
        }
        // results go to the packed output stage at once (62 accumulators would not fit the register file)
        if (c <= d1) Pf[o1 + c] = a1;
        if (c <= d2 && q != 15) Pf[o2 + c] = a2;
    }
    wave_lds_fence();
    struct __attribute__((packed, aligned(8))) d2u { double a, b; };
    d2u *dst = reinterpret_cast<d2u *>(hess + f0 * HESS_PACKED);
    for (int i = lane; i < nf * (HESS_PACKED / 2); i += 64) { d2u v; v.a = sP[2 * i]; v.b = sP[2 * i + 1]; dst[i] = v; }
}

struct PredictArgs {
    long long F;
    double grav[3];
    cpi_outputs meas;
    const double *states_i;
    long long S;
    const int *idx_i;
    double *states_j;
};
template <int MODEL>
__global__ __launch_bounds__(256) void cpi_predict_kernel(PredictArgs A) {
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= A.F) return;
    const long long ii = min(max(A.idx_i ? (long long)A.idx_i[f] : f, 0ll), A.S - 1);
    const NavState xi = ld_state(A.states_i + ii * 16);
    const NavState o = predict_state<MODEL>(xi, ldv3(A.meas.alpha + f * 3), ldv3(A.meas.beta + f * 3),
                                            ldq4(A.meas.q + f * 4), A.meas.DT[f], mk(A.grav[0], A.grav[1], A.grav[2]));
    double *d = A.states_j + f * 16;
    d[0] = o.q.x; d[1] = o.q.y; d[2] = o.q.z; d[3] = o.q.w;
    stv3(d + 4, o.bg); stv3(d + 7, o.v); stv3(d + 10, o.ba); stv3(d + 13, o.p);
}

}  // namespace

// ============================================================================================
// C-ABI
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
extern "C" const char *cpi_build_id(void) { return CPI_BUILD_ID; }

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
static const int kMeanLanes[] = {1, 2, 3, 4, 5, 6, 8, 12, 16, 32, 64};
static bool mean_lanes_supported(int L) {
    for (int c : kMeanLanes) if (c == L) return true;
    return false;
}
static int pick_lanes(const cpi_params *prm, int64_t W, int N, bool jac) {
    // model 2 with analytic Jacobians: sequential per window (the O_a / O_b recursion is not composed); model 2
    // mean-only composes through the gravity response matrices at roughly twice the arithmetic per interval
    if (prm->model == CPI_MODEL_V2 && jac) return 1;
    const double t_int = (prm->model == CPI_MODEL_V2) ? 1.2 : 0.55, t_lvl = (prm->model == CPI_MODEL_V2) ? 0.6 : 0.3;
    int L = prm->lanes_per_window;
    if (L <= 0) {
        // Small batches are latency-bound: as long as every wavefront gets a SIMD of its own (<= 1024 wavefronts
        // on MI355X) the launch lasts as long as one wavefront -- intervals per lane plus composition levels
        // (measured: ~0.55 us per interval, ~0.3 us per level).  Splitting further makes wavefronts share SIMDs
        // and loses; batches with more than 1024 single-lane wavefronts are throughput-bound and want L = 1.
        // Measured optima: L = 12 at 5 k windows x 50, 6 at 10 k, 4 at 15 k, 3 at 20 k, 2 at 30 k, 1 from ~60 k.
        double best = 1e300;
        L = 1;
        for (int c : kMeanLanes) {
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

template <int MODEL, bool JAC, bool AVG>
static void launch_mean_L(int L, const PreArgs &a, hipStream_t st) {
#define CPI_LAUNCH_L(LL)                                                                         \
    case LL: {                                                                                   \
        const long long nb = (a.W + (64 / LL) - 1) / (64 / LL);                                  \
        hipLaunchKernelGGL((cpi_mean_kernel<MODEL, JAC, AVG, LL>), dim3((unsigned)nb), dim3(64), 0, st, a); \
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
static void launch_mean(bool jac, bool avg, int L, const PreArgs &a, hipStream_t st) {
    if (jac) { if (avg) launch_mean_L<MODEL, true, true>(L, a, st); else launch_mean_L<MODEL, true, false>(L, a, st); }
    else     { if (avg) launch_mean_L<MODEL, false, true>(L, a, st); else launch_mean_L<MODEL, false, false>(L, a, st); }
}
// ---- LDS-DMA mean kernel (large mean-only batches, dense layout, one lane per window)
struct MeanDmaCfg { int kc, s, aligned; };
// CPI_AMD_MEAN_DMA = "off" | "KC,S,A" (A = 1: 16-byte aligned pieces): development override of the default below
static MeanDmaCfg mean_dma_cfg() {
    static MeanDmaCfg c = [] {
        MeanDmaCfg d = {0, 0, 0};   // default: see pick below
        const char *e = getenv("CPI_AMD_MEAN_DMA");
        if (e && strcmp(e, "off") != 0) { int k = 0, s = 0, al = 1; if (sscanf(e, "%d,%d,%d", &k, &s, &al) >= 2) d = {k, s, al}; }
        return d;
    }();
    return c;
}
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
// Returns the number of leading windows handled (a multiple of 64; the caller runs the rest through cpi_mean_kernel).
template <int MODEL>
static long long launch_mean_dma(const MeanDmaCfg &c, bool avg, const PreArgs &a, hipStream_t st) {
#define CPI_DMA_CASE(K, S_, A_) if (c.kc == K && c.s == S_ && c.aligned == A_) return launch_mean_dma_one<MODEL, K, S_, (A_ != 0)>(avg, a, st);
    CPI_DMA_CASE(4, 2, 1) CPI_DMA_CASE(4, 2, 0) CPI_DMA_CASE(2, 3, 0) CPI_DMA_CASE(8, 1, 1)
    CPI_DMA_CASE(4, 1, 0) CPI_DMA_CASE(4, 1, 1) CPI_DMA_CASE(6, 1, 0) CPI_DMA_CASE(6, 2, 0)
#undef CPI_DMA_CASE
    return 0;
}
static PreArgs shift_windows(const PreArgs &a, long long w0) {
    PreArgs t = a;
    t.W = a.W - w0;
    if (a.first) t.first = a.first + w0; else t.knots = a.knots + w0 * (long long)(a.N + 1) * 7;
    if (a.count) t.count = a.count + w0;
    t.lin = a.lin + w0 * 6;
    if (a.qk) t.qk = a.qk + w0 * 4;
    for (int k = 0; k < 12; k++) {
        static const int n[12] = { 1, 3, 3, 4, 9, 9, 9, 9, 9, 9, 9, 225 };
        double **f[12] = { &t.out.DT, &t.out.alpha, &t.out.beta, &t.out.q, &t.out.J_q, &t.out.J_a, &t.out.J_b, &t.out.H_a,
                           &t.out.H_b, &t.out.O_a, &t.out.O_b, &t.out.P };
        if (*f[k]) *f[k] += w0 * n[k];
    }
    return t;
}

// ---- block-resident mean kernel: L lanes per window (power of two) so that 64/L whole windows fit the LDS budget
// CPI_AMD_MEAN_BLK = "off" | L : development override of pick_blk_lanes
static int mean_blk_forced() {
    static int v = [] { const char *e = getenv("CPI_AMD_MEAN_BLK"); return e ? (strcmp(e, "off") == 0 ? -1 : atoi(e)) : 0; }();
    return v;
}
template <int MODEL, int L>
static void launch_mean_blk_L(bool avg, const PreArgs &a, hipStream_t st) {
    constexpr int WPB = 64 / L;
    const long long nb = (a.W + WPB - 1) / WPB;
    const size_t lds = ((size_t)WPB * (size_t)(a.N + 1) * 56 + 15) & ~(size_t)15;
    if (avg) hipLaunchKernelGGL((cpi_mean_blk_kernel<MODEL, true, L>), dim3((unsigned)nb), dim3(64), lds, st, a);
    else     hipLaunchKernelGGL((cpi_mean_blk_kernel<MODEL, false, L>), dim3((unsigned)nb), dim3(64), lds, st, a);
}
template <int MODEL>
static bool launch_mean_blk(int L, bool avg, const PreArgs &a, hipStream_t st) {
    switch (L) {
        case 8: launch_mean_blk_L<MODEL, 8>(avg, a, st); return true;
        case 16: launch_mean_blk_L<MODEL, 16>(avg, a, st); return true;
        default: return false;
    }
}

template <int MODEL>
static void launch_cov(bool avg, const PreArgs &a, hipStream_t st) {
    constexpr int G = 64 / CovDims<MODEL>::GROUP;
    const long long nb = (a.W + G - 1) / G;
    if (avg) hipLaunchKernelGGL((cpi_cov_kernel<MODEL, true>), dim3((unsigned)nb), dim3(64), 0, st, a);
    else     hipLaunchKernelGGL((cpi_cov_kernel<MODEL, false>), dim3((unsigned)nb), dim3(64), 0, st, a);
}

extern "C" int cpi_preintegrate_batch(cpi_ctx *ctx, const cpi_params *prm, int64_t W, int32_t N,
                                      const double *knots, const int64_t *first, const int32_t *count,
                                      const double *lin, const double *q_k_lin, const cpi_outputs *out) {
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
    if (L != 0 && !mean_lanes_supported(L)) return fail(ctx, CPI_ERR_INVALID, "lanes_per_window must be 0 or one of 1,2,3,4,5,6,8,12,16,32,64");

    const bool want_mean = out->DT || out->alpha || out->beta || out->q;
    const bool want_jac = out->J_q || out->J_a || out->J_b || out->H_a || out->H_b || out->O_a || out->O_b;
    const bool want_cov = out->P != nullptr;
    if (!want_mean && !want_jac && !want_cov) return CPI_OK;

    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    PreArgs a;
    memset(&a, 0, sizeof a);
    a.W = W; a.N = N; a.knots = knots; a.first = (const long long *)first; a.count = count;
    a.lin = lin; a.qk = q_k_lin;
    for (int i = 0; i < 3; i++) a.grav[i] = prm->grav[i];
    a.q4[0] = prm->sigma_w * prm->sigma_w; a.q4[1] = prm->sigma_wb * prm->sigma_wb;
    a.q4[2] = prm->sigma_a * prm->sigma_a; a.q4[3] = prm->sigma_ab * prm->sigma_ab;
    a.out = *out;
    if (prm->model == CPI_MODEL_FORSTER) {   // one kernel owns everything; imu_avg, q_k_lin, grav play no part
        hipLaunchKernelGGL(cpi_forster_kernel, dim3((unsigned)((W + 3) / 4)), dim3(64), 0, ctx->stream, a);
        CPI_HIP(ctx, hipGetLastError());
        return CPI_OK;
    }
    const bool avg = prm->imu_avg != 0;
    const bool v2 = prm->model == CPI_MODEL_V2;
    const bool stj = v2 && prm->state_transition_jacobians != 0;

    // Which kernel owns what:
    //   covariance kernel : P, and (model 2 + state_transition_jacobians) the Jacobians read out of
    //                       Discrete_J_b; it also carries the means, so it writes them when it runs.
    //   mean kernel       : means when no covariance kernel runs; the ANALYTIC Jacobians (model 1 always,
    //                       model 2 when state_transition_jacobians == 0).
    const bool run_cov = want_cov || (stj && want_jac);
    const bool mean_jac = want_jac && !stj;
    const bool run_mean = mean_jac || (want_mean && !run_cov);
    // Model 1 with Jacobians AND covariance is two independent kernels over the same knots (disjoint outputs).  The
    // covariance kernel waits on the LDS pipe about as much as it issues VALU work and uses no more than two wavefronts per
    // SIMD; the Jacobian kernel is pure FP64 VALU with no LDS: issued on a side stream (fork / join by events, so everything
    // later on the context's stream still waits for both, and a stream capture sees an ordinary fork) they share the SIMDs.
    static const bool overlap_on = [] { const char *e = getenv("CPI_AMD_NO_OVERLAP"); return !(e && atoi(e)); }();
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
        if (!want_cov) c.out.P = nullptr;
        if (v2) launch_cov<2>(avg, c, ctx->stream); else launch_cov<1>(avg, c, ctx->stream);
    }
    if (run_mean) {
        PreArgs m = a;
        m.write_means = (want_mean && !run_cov) ? 1 : 0;
        m.write_jac = mean_jac ? 1 : 0;
        const int LL = pick_lanes(prm, W, N, mean_jac);
        long long done = 0;
        const int bl = mean_blk_forced();
        static const int dbg_mode = [] { const char *e = getenv("CPI_AMD_BLK_MODE"); return e ? atoi(e) : 0; }();   // development only
        m.dbg = dbg_mode;
        if (!mean_jac && !first && bl > 0 && (size_t)(64 / bl) * (size_t)(N + 1) * 56 <= 65536 && N >= 1) {
            if (v2 ? launch_mean_blk<2>(bl, avg, m, ctx->stream) : launch_mean_blk<1>(bl, avg, m, ctx->stream)) done = W;
        }
        const MeanDmaCfg dc = mean_dma_cfg();
        if (!done && !mean_jac && LL == 1 && !first && !count && dc.kc > 0 && N >= 2 * dc.kc)
            done = v2 ? launch_mean_dma<2>(dc, avg, m, ctx->stream) : launch_mean_dma<1>(dc, avg, m, ctx->stream);
        if (done < W) {
            const PreArgs t = done ? shift_windows(m, done) : m;
            if (v2) launch_mean<2>(mean_jac, avg, LL, t, mean_stream); else launch_mean<1>(mean_jac, avg, LL, t, mean_stream);
        }
    }
    if (forked) {
        CPI_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->side));
        CPI_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    }
    CPI_HIP(ctx, hipGetLastError());
    return CPI_OK;
}

// Lanes per factor of the factor kernel.  CPI_AMD_FACTOR_LANES (16 | 8 | 4) overrides the size heuristic
// (tuning / A-B measurements).
static int factor_lanes(int64_t F, bool whiten) {
    static int forced = [] {
        const char *e = getenv("CPI_AMD_FACTOR_LANES");
        const int v = e ? atoi(e) : 0;
        return (v == 16 || v == 8 || v == 4) ? v : 0;
    }();
    if (forced) return forced;
    // measured on MI355X, 1 M factors: plain 0.82 ms with 8 lanes vs 1.09 ms with 16; whitened (37 KB vs 19 KB of
    // LDS per wavefront) 1.45 ms with 8 vs 1.37 ms with 16
    if (whiten) return 16;
    return F >= 32768 ? 8 : 16;
}

static int factor_eval_impl(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F, const cpi_outputs *meas,
                            const double *lin, const double *q_k_lin, const double *states, int64_t S, const int32_t *idx_i,
                            const int32_t *idx_j, const double *sqrt_info, double *err, double *H1, double *H2);

extern "C" int cpi_factor_eval_batch(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                                     const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                                     const double *states, int64_t S, const int32_t *idx_i, const int32_t *idx_j,
                                     double *err, double *H1, double *H2) {
    return factor_eval_impl(ctx, model, grav, F, meas, lin, q_k_lin, states, S, idx_i, idx_j, nullptr, err, H1, H2);
}

static int factor_eval_impl(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F, const cpi_outputs *meas,
                            const double *lin, const double *q_k_lin, const double *states, int64_t S, const int32_t *idx_i,
                            const int32_t *idx_j, const double *sqrt_info, double *err, double *H1, double *H2) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (model != CPI_MODEL_V1 && model != CPI_MODEL_V2) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_batch: model must be 1 or 2");
    if (F < 0) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_batch: negative size");
    if (F == 0) return CPI_OK;
    if (!grav || !meas || !lin || !states || !err) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_batch: NULL argument");
    if (!meas->DT || !meas->alpha || !meas->beta || !meas->q || !meas->J_q || !meas->J_a || !meas->J_b || !meas->H_a || !meas->H_b)
        return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_batch: measurement fields DT/alpha/beta/q/J_q/J_a/J_b/H_a/H_b are required");
    if (model == CPI_MODEL_V2 && (!q_k_lin || !meas->O_a || !meas->O_b))
        return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_batch: model 2 needs q_k_lin, O_a, O_b");
    if (!grid_ok(F)) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_batch: F exceeds 2^31 - 1 factors per call");
    if (S <= 0 || (!idx_i && S < F) || (!idx_j && S < F + 1))
        return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_batch: S (number of states) must be >= 1, and >= F + 1 when idx_i / idx_j are NULL (chained states f, f + 1)");
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    FactorArgs a;
    memset(&a, 0, sizeof a);
    a.F = F;
    for (int i = 0; i < 3; i++) a.grav[i] = grav[i];
    a.meas = *meas; a.lin = lin; a.qk = q_k_lin; a.states = states; a.S = S; a.idx_i = idx_i; a.idx_j = idx_j;
    a.err = err; a.H1 = H1; a.H2 = H2; a.sqrt_info = sqrt_info;
    // lanes per factor: 16 gives the most wavefronts (small sweeps), fewer lanes do less redundant arithmetic
    const int lpf = factor_lanes(F, sqrt_info != nullptr);
#define CPI_LAUNCH_FACTOR(M, WH, L) \
    hipLaunchKernelGGL((cpi_factor_kernel<M, WH, L>), dim3((unsigned)((F + 64 / L - 1) / (64 / L))), dim3(64), 0, ctx->stream, a)
#define CPI_LAUNCH_FACTOR_L(M, WH) \
    do { if (lpf == 16) CPI_LAUNCH_FACTOR(M, WH, 16); else if (lpf == 8) CPI_LAUNCH_FACTOR(M, WH, 8); else CPI_LAUNCH_FACTOR(M, WH, 4); } while (0)
    if (sqrt_info) {
        if (model == CPI_MODEL_V1) CPI_LAUNCH_FACTOR_L(1, true); else CPI_LAUNCH_FACTOR_L(2, true);
    } else {
        if (model == CPI_MODEL_V1) CPI_LAUNCH_FACTOR_L(1, false); else CPI_LAUNCH_FACTOR_L(2, false);
    }
#undef CPI_LAUNCH_FACTOR_L
#undef CPI_LAUNCH_FACTOR
    CPI_HIP(ctx, hipGetLastError());
    return CPI_OK;
}

extern "C" int cpi_factor_eval_packed_batch(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                                            const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                                            const double *states, int64_t S, const int32_t *idx_i, const int32_t *idx_j,
                                            double *packed) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (model != CPI_MODEL_V1 && model != CPI_MODEL_V2) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_packed_batch: model must be 1 or 2");
    if (F < 0) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_packed_batch: negative size");
    if (F == 0) return CPI_OK;
    if (!grav || !meas || !lin || !states || !packed) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_packed_batch: NULL argument");
    if (!meas->DT || !meas->alpha || !meas->beta || !meas->q || !meas->J_q || !meas->J_a || !meas->J_b || !meas->H_a || !meas->H_b)
        return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_packed_batch: measurement fields DT/alpha/beta/q/J_q/J_a/J_b/H_a/H_b are required");
    if (model == CPI_MODEL_V2 && (!q_k_lin || !meas->O_a || !meas->O_b))
        return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_packed_batch: model 2 needs q_k_lin, O_a, O_b");
    if (!grid_ok(F)) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_packed_batch: F exceeds 2^31 - 1 factors per call");
    if (S <= 0 || (!idx_i && S < F) || (!idx_j && S < F + 1))
        return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_packed_batch: S (number of states) must be >= 1, and >= F + 1 when idx_i / idx_j are NULL");
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    FactorArgs a;
    memset(&a, 0, sizeof a);
    a.F = F;
    for (int i = 0; i < 3; i++) a.grav[i] = grav[i];
    a.meas = *meas; a.lin = lin; a.qk = q_k_lin; a.states = states; a.S = S; a.idx_i = idx_i; a.idx_j = idx_j;
    // lanes per factor.  Every lane of a factor repeats the shared quaternion algebra, so fewer lanes = less VALU per factor
    // but more LDS per wavefront (a factor's staged record + packed output = 1.5 KB).  Measured (MI355X, 1 M factors, model 1 /
    // model 2, us): 8 lanes 383 / 435 (VALU 53 % busy at 2 wavefronts per SIMD), 6: 335 / 389, 4: 290 / 324, 3: 281 / 314,
    // 2: 328 / 361 (48 KB of LDS: one wavefront per SIMD); 100 k factors: 4 lanes 31.5, 3 lanes 32.5.
    int lpf = (F >= 300000) ? 3 : 4;
    if (const char *e = getenv("CPI_AMD_PACKED_LPF")) lpf = atoi(e);   // measurements
#define CPI_PACKED(M, L) hipLaunchKernelGGL((cpi_factor_packed_kernel<M, L>), dim3((unsigned)((F + 64 / L - 1) / (64 / L))), dim3(64), 0, ctx->stream, a, packed)
    if (model == CPI_MODEL_V1) { if (lpf == 2) CPI_PACKED(1, 2); else if (lpf == 3) CPI_PACKED(1, 3); else if (lpf == 4) CPI_PACKED(1, 4); else if (lpf == 6) CPI_PACKED(1, 6); else CPI_PACKED(1, 8); }
    else                       { if (lpf == 2) CPI_PACKED(2, 2); else if (lpf == 3) CPI_PACKED(2, 3); else if (lpf == 4) CPI_PACKED(2, 4); else if (lpf == 6) CPI_PACKED(2, 6); else CPI_PACKED(2, 8); }
#undef CPI_PACKED
    CPI_HIP(ctx, hipGetLastError());
    return CPI_OK;
}

extern "C" int cpi_sqrt_information_batch(cpi_ctx *ctx, int64_t F, const double *P, double *sqrt_info) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (F < 0) return fail(ctx, CPI_ERR_INVALID, "cpi_sqrt_information_batch: negative size");
    if (F == 0) return CPI_OK;
    if (!P || !sqrt_info) return fail(ctx, CPI_ERR_INVALID, "cpi_sqrt_information_batch: NULL argument");
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    const long long nb = (F + 3) / 4;
    hipLaunchKernelGGL(cpi_sqrt_info_kernel, dim3((unsigned)nb), dim3(64), 0, ctx->stream, (long long)F, P, sqrt_info);
    CPI_HIP(ctx, hipGetLastError());
    return CPI_OK;
}

extern "C" int cpi_factor_eval_whitened_batch(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                                              const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                                              const double *states, int64_t S, const int32_t *idx_i, const int32_t *idx_j,
                                              const double *sqrt_info, double *err, double *H1, double *H2) {
    if (ctx && !sqrt_info) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_eval_whitened_batch: sqrt_info is NULL");
    return factor_eval_impl(ctx, model, grav, F, meas, lin, q_k_lin, states, S, idx_i, idx_j, sqrt_info, err, H1, H2);
}

extern "C" int cpi_factor_hessian_batch(cpi_ctx *ctx, int32_t model, const double grav[3], int64_t F,
                                        const cpi_outputs *meas, const double *lin, const double *q_k_lin,
                                        const double *states, int64_t S, const int32_t *idx_i, const int32_t *idx_j,
                                        const double *sqrt_info, double *hess) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (model != CPI_MODEL_V1 && model != CPI_MODEL_V2) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_hessian_batch: model must be 1 or 2");
    if (F < 0) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_hessian_batch: negative size");
    if (F == 0) return CPI_OK;
    if (!grav || !meas || !lin || !states || !sqrt_info || !hess) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_hessian_batch: NULL argument");
    if (!meas->DT || !meas->alpha || !meas->beta || !meas->q || !meas->J_q || !meas->J_a || !meas->J_b || !meas->H_a || !meas->H_b)
        return fail(ctx, CPI_ERR_INVALID, "cpi_factor_hessian_batch: measurement fields DT/alpha/beta/q/J_q/J_a/J_b/H_a/H_b are required");
    if (model == CPI_MODEL_V2 && (!q_k_lin || !meas->O_a || !meas->O_b))
        return fail(ctx, CPI_ERR_INVALID, "cpi_factor_hessian_batch: model 2 needs q_k_lin, O_a, O_b");
    if (!grid_ok(F)) return fail(ctx, CPI_ERR_INVALID, "cpi_factor_hessian_batch: F exceeds 2^31 - 1 factors per call");
    if (S <= 0 || (!idx_i && S < F) || (!idx_j && S < F + 1))
        return fail(ctx, CPI_ERR_INVALID, "cpi_factor_hessian_batch: S (number of states) must be >= 1, and >= F + 1 when idx_i / idx_j are NULL");
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    FactorArgs a;
    memset(&a, 0, sizeof a);
    a.F = F;
    for (int i = 0; i < 3; i++) a.grav[i] = grav[i];
    a.meas = *meas; a.lin = lin; a.qk = q_k_lin; a.states = states; a.S = S; a.idx_i = idx_i; a.idx_j = idx_j; a.sqrt_info = sqrt_info;
    const unsigned nb = (unsigned)((F + 3) / 4);
    if (model == CPI_MODEL_V1) hipLaunchKernelGGL((cpi_factor_hessian_kernel<1>), dim3(nb), dim3(64), 0, ctx->stream, a, hess);
    else hipLaunchKernelGGL((cpi_factor_hessian_kernel<2>), dim3(nb), dim3(64), 0, ctx->stream, a, hess);
    CPI_HIP(ctx, hipGetLastError());
    return CPI_OK;
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
    const long long nb = (F + 255) / 256;
    if (model == CPI_MODEL_V1) hipLaunchKernelGGL((cpi_predict_kernel<1>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL((cpi_predict_kernel<2>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, a);
    CPI_HIP(ctx, hipGetLastError());
    return CPI_OK;
}

static const int OUT_N[12] = { 1, 3, 3, 4, 9, 9, 9, 9, 9, 9, 9, 225 };
static double **out_field(cpi_outputs *o, int k) {
    double **f[12] = { &o->DT, &o->alpha, &o->beta, &o->q, &o->J_q, &o->J_a, &o->J_b, &o->H_a, &o->H_b, &o->O_a, &o->O_b, &o->P };
    return f[k];
}

// -------------------------------------------------------------------------------- tiled layout
extern "C" int cpi_tile_knots(cpi_ctx *ctx, int64_t W, int32_t N, const double *knots, double *tiles) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (W < 0 || N < 0) return fail(ctx, CPI_ERR_INVALID, "cpi_tile_knots: negative size");
    if (W == 0) return CPI_OK;
    if (!knots || !tiles) return fail(ctx, CPI_ERR_INVALID, "cpi_tile_knots: NULL argument");
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    const long long total = ((W + 63) / 64) * (long long)(N + 1) * 448;
    const unsigned nb = (unsigned)std::min<long long>((total + 255) / 256, 256 * 64);
    // tile-major.  (Step-major -- tiles[N+1][ceil(W/64)][7][64], all resident wavefronts reading one moving window -- was
    // measured: 557 vs 571 us per 1 M x 50, 58.2 vs 61.1 us per 100 k, stream alone 472 vs 481 us: not worth a second contract.)
    const long long ts = (long long)(N + 1) * 448, ss = 448;
    hipLaunchKernelGGL(cpi_tile_knots_kernel, dim3(nb), dim3(256), 0, ctx->stream, (long long)W, (int)N, knots, tiles, ts, ss);
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
    if (out->J_q || out->J_a || out->J_b || out->H_a || out->H_b || out->O_a || out->O_b || out->P)
        return fail(ctx, CPI_ERR_INVALID, "cpi_preintegrate_tiled_batch: the tiled layout serves the mean outputs (DT, alpha, beta, q) only; "
                                          "Jacobians and covariance are FP64-bound, not HBM-bound: use cpi_preintegrate_batch");
    if (!(out->DT || out->alpha || out->beta || out->q)) return CPI_OK;
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    TiledArgs a;
    memset(&a, 0, sizeof a);
    a.W = W; a.N = N; a.tiles = tiles; a.count = count; a.lin = lin; a.qk = q_k_lin; a.out = *out;
    for (int i = 0; i < 3; i++) a.grav[i] = prm->grav[i];
    if (const char *e = getenv("CPI_AMD_BLK_MODE")) a.dbg = atoi(e);
    a.ts = (long long)(N + 1) * 448; a.ss = 448;
    const unsigned nb = (unsigned)((W + 63) / 64);
    const bool avg = prm->imu_avg != 0;
    if (a.dbg == 1) {   // CPI_AMD_PROBE_LDS = dynamic LDS bytes per wavefront, to pin the probe's occupancy (13312 -> 12 waves / CU)
        const char *e = getenv("CPI_AMD_PROBE_LDS");
        hipLaunchKernelGGL(cpi_tiled_fetch_probe_kernel, dim3(nb), dim3(64), e ? atoi(e) : 0, ctx->stream, a);
        return CPI_OK;
    }
    // wavefronts per tile.  Measured (MI355X, N = 50, us per launch, S = 1 / 2 / 3 / 4 / 8): 5 k windows 24.9 / 15.0 / 11.5 /
    // 10.3 / -, 10 k 25.2 / 15.4 / 11.9 / 10.8 / 12.4, 20 k 27.0 / 23.6 / 19.6 / 18.7 / 22.8, 30 k 28.6 / 25.1 / 22.2 / 21.3,
    // 50 k 33.0 / 34.4 / 34.8 / 34.9, 100 k 61.3 / 64.3 / 65.8 / 65.2: four (one per SIMD of the CU that owns the tile) while
    // the tiles do not fill the chip, one beyond.  CPI_AMD_TILED_SPLIT overrides (measurements, tests).
    int S = (nb < 640) ? std::max(1, std::min(4, (int)N / 4)) : 1;
    if (const char *e = getenv("CPI_AMD_TILED_SPLIT")) S = std::max(1, std::min(8, atoi(e)));
    const size_t lds = (size_t)(S - 1) * (prm->model == CPI_MODEL_V2 ? 34 : 16) * 64 * sizeof(double);
    // more than 64 KB of dynamic LDS (model 2, S >= 5) needs the kernel's limit raised once per device
#define CPI_TILED2(M, AV, C)                                                                                      \
    do {                                                                                                          \
        if (S > 1) {                                                                                              \
            const unsigned bit = 1u << ((M - 1) * 4 + (AV ? 2 : 0) + (C ? 1 : 0));                                \
            if (lds > 65536 && !(ctx->big_lds_set & bit)) {                                                       \
                CPI_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&cpi_mean_tiled_kernel<M, AV, C, true>), \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 7 * 34 * 64 * 8));   \
                ctx->big_lds_set |= bit;                                                                          \
            }                                                                                                     \
            hipLaunchKernelGGL((cpi_mean_tiled_kernel<M, AV, C, true>), dim3(nb), dim3(64 * S), lds, ctx->stream, a); \
        } else hipLaunchKernelGGL((cpi_mean_tiled_kernel<M, AV, C, false>), dim3(nb), dim3(64), 0, ctx->stream, a); \
    } while (0)
    if (prm->model == CPI_MODEL_V1) {
        if (count) { if (avg) CPI_TILED2(1, true, true); else CPI_TILED2(1, false, true); }
        else       { if (avg) CPI_TILED2(1, true, false); else CPI_TILED2(1, false, false); }
    } else {
        if (count) { if (avg) CPI_TILED2(2, true, true); else CPI_TILED2(2, false, true); }
        else       { if (avg) CPI_TILED2(2, true, false); else CPI_TILED2(2, false, false); }
    }
#undef CPI_TILED2
    CPI_HIP(ctx, hipGetLastError());
    return CPI_OK;
}

// -------------------------------------------------------------------------------- device sets
// Windows shard embarrassingly: rank r of n owns the contiguous block cpi_shard_bounds(W, r, n) and runs the ordinary
// entries on its own context; the ONE exchange step is the final gather of the output slabs to a root device.  RCCL is
// bound lazily (dlopen of librccl.so.1 at the first cpi_group_create): single-GPU users never load it, and a process
// that already carries an RCCL (PyTorch) shares that copy.  One process drives all devices (ncclCommInitAll,
// rccl/rccl.h:236) -- the reference is a single process too; multi-process hosts (one rank per GPU, torch.distributed)
// use cpi_amd/dist.py, which issues the same send / recv pattern through ProcessGroupNCCL.
#include <dlfcn.h>
#include <mutex>
#include <vector>
namespace {
struct Rccl {
    void *h = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
    std::mutex mu;
    bool load() {   // serialised: two host threads may create their first groups at the same time
        std::lock_guard<std::mutex> lock(mu);
        if (h) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) { err = std::string("dlopen(librccl.so.1): ") + (dlerror() ? dlerror() : "not found"); return false; }
#define CPI_SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(h, name)); if (!field) { err = std::string("dlsym ") + name; h = nullptr; return false; }
        CPI_SYM(CommInitAll, "ncclCommInitAll") CPI_SYM(CommDestroy, "ncclCommDestroy") CPI_SYM(GroupStart, "ncclGroupStart")
        CPI_SYM(GroupEnd, "ncclGroupEnd") CPI_SYM(Send, "ncclSend") CPI_SYM(Recv, "ncclRecv") CPI_SYM(GetErrorString, "ncclGetErrorString")
#undef CPI_SYM
        return true;
    }
};
Rccl g_rccl;
constexpr int kNcclDouble = 8;   // ncclFloat64 (rccl/rccl.h:467)
}  // namespace

struct cpi_group {
    int n = 0;
    std::vector<cpi_ctx *> ctx;
    std::vector<hipStream_t> streams;   // owned
    std::vector<void *> comms;          // ncclComm_t, empty when n == 1
    std::string err;
};
static thread_local std::string g_group_err;
static int gfail(cpi_group *g, int code, const std::string &msg) { if (g) g->err = msg; else g_group_err = msg; return code; }

extern "C" void cpi_shard_bounds(int64_t W, int rank, int n, int64_t *lo, int64_t *hi) {
    const int64_t per = n > 0 ? (W + n - 1) / n : W;
    const int64_t a = std::min<int64_t>(W, (int64_t)rank * per);
    if (lo) *lo = a;
    if (hi) *hi = std::min<int64_t>(W, a + per);
}
extern "C" const char *cpi_group_last_error(const cpi_group *g) { return g ? g->err.c_str() : g_group_err.c_str(); }
extern "C" int cpi_group_size(const cpi_group *g) { return g ? g->n : 0; }
extern "C" cpi_ctx *cpi_group_ctx(cpi_group *g, int rank) { return (g && rank >= 0 && rank < g->n) ? g->ctx[rank] : nullptr; }
extern "C" void cpi_group_destroy(cpi_group *g) {
    if (!g) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (int r = 0; r < g->n; r++) {
        if (r < (int)g->comms.size() && g->comms[r] && g_rccl.CommDestroy) g_rccl.CommDestroy(g->comms[r]);
        if (r < (int)g->ctx.size() && g->ctx[r]) {
            (void)hipSetDevice(g->ctx[r]->device);
            if (r < (int)g->streams.size() && g->streams[r]) (void)hipStreamDestroy(g->streams[r]);
            cpi_ctx_destroy(g->ctx[r]);
        }
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    delete g;
}
extern "C" int cpi_group_create(int n, const int *devices, cpi_group **out) {
    if (!out) return gfail(nullptr, CPI_ERR_INVALID, "cpi_group_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return gfail(nullptr, CPI_ERR_NO_DEVICE, "cpi_group_create: no HIP device available (this library has no CPU fallback)");
    if (n <= 0 || n > ndev) return gfail(nullptr, CPI_ERR_INVALID, "cpi_group_create: n must be between 1 and the number of devices");
    std::vector<int> devs(n);
    for (int r = 0; r < n; r++) {
        devs[r] = devices ? devices[r] : r;
        if (devs[r] < 0 || devs[r] >= ndev) return gfail(nullptr, CPI_ERR_INVALID, "cpi_group_create: device index out of range");
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
        const int rc = g_rccl.CommInitAll(g->comms.data(), n, devs.data());
        if (rc != 0) { const std::string m = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(rc); cpi_group_destroy(g); return gfail(nullptr, CPI_ERR_RCCL, m); }
    }
    *out = g;
    return CPI_OK;
}
extern "C" int cpi_group_synchronize(cpi_group *g) {
    if (!g) return gfail(nullptr, CPI_ERR_INVALID, "group is NULL");
    for (int r = 0; r < g->n; r++) {
        const int rc = cpi_ctx_synchronize(g->ctx[r]);
        if (rc != CPI_OK) return gfail(g, rc, cpi_last_error(g->ctx[r]));
    }
    return CPI_OK;
}
extern "C" int cpi_group_gather(cpi_group *g, int root, int64_t W, const cpi_outputs *local, const cpi_outputs *root_out) {
    if (!g) return gfail(nullptr, CPI_ERR_INVALID, "group is NULL");
    if (root < 0 || root >= g->n || W < 0 || !local || !root_out) return gfail(g, CPI_ERR_INVALID, "cpi_group_gather: invalid argument");
    int prev = -1;
    (void)hipGetDevice(&prev);
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore_{prev};
    int rc = 0;
    if (g->n > 1) { rc = g_rccl.GroupStart(); if (rc) return gfail(g, CPI_ERR_RCCL, std::string("ncclGroupStart: ") + g_rccl.GetErrorString(rc)); }
    for (int k = 0; k < 12 && rc == 0; k++) {
        cpi_outputs ro = *root_out;
        double *dst = *out_field(&ro, k);
        if (!dst) continue;
        for (int r = 0; r < g->n && rc == 0; r++) {
            int64_t lo, hi;
            cpi_shard_bounds(W, r, g->n, &lo, &hi);
            const size_t cnt = (size_t)(hi - lo) * (size_t)OUT_N[k];
            if (cnt == 0) continue;
            cpi_outputs lr = local[r];
            const double *src = *out_field(&lr, k);
            if (!src) { rc = -1; break; }
            if (r == root) {   // the root's own block: device-to-device copy on its stream (unless it was computed in place)
                if (src != dst + (size_t)lo * OUT_N[k]) {
                    if (hipSetDevice(g->ctx[root]->device) != hipSuccess ||
                        hipMemcpyAsync(dst + (size_t)lo * OUT_N[k], src, cnt * sizeof(double), hipMemcpyDeviceToDevice, g->streams[root]) != hipSuccess) rc = -2;
                }
            } else {           // every peer sends its slab straight to the root: one xGMI link per peer, no ring
                rc = g_rccl.Recv(dst + (size_t)lo * OUT_N[k], cnt, kNcclDouble, r, g->comms[root], g->streams[root]);
                if (rc == 0) rc = g_rccl.Send(src, cnt, kNcclDouble, root, g->comms[r], g->streams[r]);
            }
        }
    }
    int rce = 0;
    if (g->n > 1) rce = g_rccl.GroupEnd();
    if (rc == -1) return gfail(g, CPI_ERR_INVALID, "cpi_group_gather: a field wanted at the root is NULL in a rank's local outputs");
    if (rc == -2) return gfail(g, CPI_ERR_HIP, "cpi_group_gather: device-to-device copy of the root's own block failed");
    if (rc) return gfail(g, CPI_ERR_RCCL, std::string("ncclSend/ncclRecv: ") + g_rccl.GetErrorString(rc));
    if (rce) return gfail(g, CPI_ERR_RCCL, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(rce));
    return CPI_OK;
}

// -------------------------------------------------------------------------------- test hook (include/cpi_amd_test.h)
namespace {
__global__ __launch_bounds__(64) void cpi_test_quat_ops_kernel(int op, long long n, const double *in, double *out) {
    const long long k = (long long)blockIdx.x * 64 + threadIdx.x;
    if (k >= n) return;
    auto put_rm = [&](double *o, const M3 &A) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) o[i * 3 + j] = A.m[i][j];
    };
    auto put_q = [&](double *o, Q4 q) { o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w; };
    switch (op) {
        case 0: put_q(out + 4 * k, rot_2_quat(rec_mat(in + 9 * k, 0))); break;
        case 1: put_rm(out + 9 * k, skew(ldv3(in + 3 * k))); break;
        case 2: put_rm(out + 9 * k, quat_2_Rot(ldq4(in + 4 * k))); break;
        case 3: put_q(out + 4 * k, quat_multiply(ldq4(in + 8 * k), ldq4(in + 8 * k + 4))); break;
        case 4: put_rm(out + 9 * k, Exp_so3(ldv3(in + 3 * k))); break;
        default: put_q(out + 4 * k, quat_inv(ldq4(in + 4 * k))); break;
    }
}
}  // namespace
extern "C" int cpi_test_quat_ops(cpi_ctx *ctx, int32_t op, int64_t n, const double *in, double *out) {
    if (!ctx) return fail(nullptr, CPI_ERR_INVALID, "ctx is NULL");
    if (op < 0 || op > 5 || n < 0) return fail(ctx, CPI_ERR_INVALID, "cpi_test_quat_ops: unknown op / negative size");
    if (n == 0) return CPI_OK;
    if (!in || !out) return fail(ctx, CPI_ERR_INVALID, "cpi_test_quat_ops: NULL argument");
    DeviceGuard guard_;
    CPI_HIP(ctx, guard_.enter(ctx->device));
    hipLaunchKernelGGL(cpi_test_quat_ops_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, (int)op, (long long)n, in, out);
    CPI_HIP(ctx, hipGetLastError());
    return CPI_OK;
}

// -------------------------------------------------------------------------------- host-pointer variants
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

// Dense batches from host memory run as a three-stage pipeline over chunks of <= 65536 windows: upload of chunk i + 1
// (copy stream), kernels of chunk i (the context's stream), download of chunk i - 1 (second copy stream) -- PCIe is full
// duplex, so with PINNED host buffers (cpi_host_alloc, hipHostMalloc, torch pin_memory) a call costs about
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
    void *out[2][12] = {};
    size_t out_cap[2][12] = {};
};
static void host_pipe_destroy(HostPipe *hp) {
    if (!hp) return;
    for (int s = 0; s < 2; s++) {
        for (int k = 0; k < 4; k++) if (hp->in[s][k]) (void)hipFree(hp->in[s][k]);
        for (int k = 0; k < 12; k++) if (hp->out[s][k]) (void)hipFree(hp->out[s][k]);
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

static int preintegrate_host_dense(cpi_ctx *ctx, const cpi_params *prm, int64_t W, int32_t N, const double *knots,
                                   const int32_t *count, const double *lin, const double *q_k_lin, const cpi_outputs *out) {
    int rc = host_pipe_get(ctx);
    if (rc != CPI_OK) return rc;
    HostPipe *hp = ctx->pipe;
    const int64_t nch = (W + 65535) / 65536;
    const int64_t Wc = std::min<int64_t>(W, (((W + nch - 1) / nch) + 63) / 64 * 64);   // balanced chunks, whole wavefronts
    const int nslots = nch > 1 ? 2 : 1;
    const size_t knot_bytes = (size_t)(N + 1) * 7 * sizeof(double);
    cpi_outputs h = *out;
    for (int s = 0; s < nslots; s++) {
        const size_t need[4] = { (size_t)Wc * knot_bytes, count ? (size_t)Wc * sizeof(int32_t) : 0, (size_t)Wc * 6 * sizeof(double),
                                 q_k_lin ? (size_t)Wc * 4 * sizeof(double) : 0 };
        for (int k = 0; k < 4; k++)
            if ((rc = host_pipe_reserve(ctx, hp->in[s][k], hp->in_cap[s][k], need[k])) != CPI_OK) return rc;
        for (int k = 0; k < 12; k++)
            if (*out_field(&h, k) && (rc = host_pipe_reserve(ctx, hp->out[s][k], hp->out_cap[s][k], (size_t)Wc * OUT_N[k] * sizeof(double))) != CPI_OK) return rc;
    }
    // after the first enqueue nothing may return before the three streams are idle: copies into the caller's memory are in flight
    std::string err;
    auto hip_ok = [&](hipError_t e, const char *what) { if (e != hipSuccess && err.empty()) err = std::string(what) + ": " + hipGetErrorString(e); return e == hipSuccess; };
    for (int64_t i = 0; i < nch && err.empty() && rc == CPI_OK; i++) {
        const int s = (int)(i & 1);
        const int64_t w0 = i * Wc, wn = std::min<int64_t>(Wc, W - w0);
        if (i >= 2 && !hip_ok(hipStreamWaitEvent(hp->up, hp->ev_done[s], 0), "hipStreamWaitEvent")) break;   // slot's inputs consumed
        if (!hip_ok(hipMemcpyAsync(hp->in[s][0], knots + (size_t)w0 * (N + 1) * 7, (size_t)wn * knot_bytes, hipMemcpyHostToDevice, hp->up), "upload knots")) break;
        if (count && !hip_ok(hipMemcpyAsync(hp->in[s][1], count + w0, (size_t)wn * sizeof(int32_t), hipMemcpyHostToDevice, hp->up), "upload count")) break;
        if (!hip_ok(hipMemcpyAsync(hp->in[s][2], lin + (size_t)w0 * 6, (size_t)wn * 6 * sizeof(double), hipMemcpyHostToDevice, hp->up), "upload lin")) break;
        if (q_k_lin && !hip_ok(hipMemcpyAsync(hp->in[s][3], q_k_lin + (size_t)w0 * 4, (size_t)wn * 4 * sizeof(double), hipMemcpyHostToDevice, hp->up), "upload q_k_lin")) break;
        if (!hip_ok(hipEventRecord(hp->ev_in[s], hp->up), "hipEventRecord")) break;
        if (!hip_ok(hipStreamWaitEvent(ctx->stream, hp->ev_in[s], 0), "hipStreamWaitEvent")) break;
        if (i >= 2 && !hip_ok(hipStreamWaitEvent(ctx->stream, hp->ev_out[s], 0), "hipStreamWaitEvent")) break;   // slot's outputs downloaded
        cpi_outputs d;
        memset(&d, 0, sizeof d);
        for (int k = 0; k < 12; k++) if (*out_field(&h, k)) *out_field(&d, k) = (double *)hp->out[s][k];
        rc = cpi_preintegrate_batch(ctx, prm, wn, N, (const double *)hp->in[s][0], nullptr, count ? (const int32_t *)hp->in[s][1] : nullptr,
                                    (const double *)hp->in[s][2], q_k_lin ? (const double *)hp->in[s][3] : nullptr, &d);
        if (rc != CPI_OK) break;
        if (!hip_ok(hipEventRecord(hp->ev_done[s], ctx->stream), "hipEventRecord")) break;
        if (!hip_ok(hipStreamWaitEvent(hp->down, hp->ev_done[s], 0), "hipStreamWaitEvent")) break;
        for (int k = 0; k < 12; k++)
            if (*out_field(&h, k) && !hip_ok(hipMemcpyAsync(*out_field(&h, k) + (size_t)w0 * OUT_N[k], hp->out[s][k], (size_t)wn * OUT_N[k] * sizeof(double),
                                                            hipMemcpyDeviceToHost, hp->down), "download")) break;
        if (!err.empty()) break;
        if (!hip_ok(hipEventRecord(hp->ev_out[s], hp->down), "hipEventRecord")) break;
    }
    const hipError_t e1 = hipStreamSynchronize(hp->up), e2 = hipStreamSynchronize(ctx->stream), e3 = hipStreamSynchronize(hp->down);
    if (rc != CPI_OK) return rc;   // message already set by cpi_preintegrate_batch
    if (!err.empty()) return fail(ctx, CPI_ERR_HIP, "cpi_preintegrate_batch_host: " + err);
    hip_ok(e1, "hipStreamSynchronize(upload)"); hip_ok(e2, "hipStreamSynchronize"); hip_ok(e3, "hipStreamSynchronize(download)");
    if (!err.empty()) return fail(ctx, CPI_ERR_HIP, "cpi_preintegrate_batch_host: " + err);
    return CPI_OK;
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
    if (!first) return preintegrate_host_dense(ctx, prm, W, N, knots, count, lin, q_k_lin, out);
    // ragged windows share one knot stream: staged whole (one-off calls; the stream is usually small)
    DevBuf dk, df, dc, dl, dq, dout[12];
    CPI_UP(dk, knots, (size_t)n_knots * 7 * sizeof(double));
    CPI_UP(df, first, (size_t)W * sizeof(int64_t));
    CPI_UP(dc, count, (size_t)W * sizeof(int32_t));
    CPI_UP(dl, lin, (size_t)W * 6 * sizeof(double));
    CPI_UP(dq, q_k_lin, (size_t)W * 4 * sizeof(double));
    cpi_outputs d = *out, h = *out;
    for (int k = 0; k < 12; k++)
        if (*out_field(&h, k)) {
            CPI_HIP(ctx, hipMalloc(&dout[k].p, (size_t)W * OUT_N[k] * sizeof(double)));
            *out_field(&d, k) = (double *)dout[k].p;
        }
    int rc = cpi_preintegrate_batch(ctx, prm, W, N, (const double *)dk.p, (const int64_t *)df.p, (const int32_t *)dc.p,
                                    (const double *)dl.p, (const double *)dq.p, &d);
    if (rc != CPI_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    for (int k = 0; k < 12; k++)
        if (*out_field(&h, k))
            CPI_HIP(ctx, hipMemcpyAsync(*out_field(&h, k), dout[k].p, (size_t)W * OUT_N[k] * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
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
