// cpi_math.hpp -- per-lane arithmetic of the MI355X continuous-preintegration kernels.
//
// Everything here is straight-line f64 code on 3-vectors / 3x3 blocks that lives in VGPRs: the
// contractions are far too small for MFMA.  The kernels (cpi_mean / cpi_cov / cpi_factor .hip) own the lane mapping,
// LDS staging and cross-lane exchange; this header owns the mathematics so that it can also be
// compiled for the host by tests/hostsim (a lane-by-lane emulator used ONLY by the CPU test suite
// to validate kernel logic where no GPU is available -- it is not reachable from the C-ABI).
//
// What is computed (reference file:line):
//   per-sample closed forms        cpi_compare/src/cpi/CpiV1.h:67-154 (means), :161-259 (analytic
//                                  bias Jacobians); CpiV2.h:88-305 (model 2 incl. O_a/O_b)
//   covariance / state transition  CpiV1.h:266-353 (15x15 RK4), CpiV2.h:314-464 (21x21 RK4 + Phi,
//                                  clone / marginalise, Jacobian read-out)
//   SO(3) / JPL quaternion helpers cpi_compare/src/utils/quat_ops.h:45-197
//   factor residual + Jacobians    cpi_compare/src/gtsam/ImuFactorCPIv1.cpp:37-208, v2.cpp:38-212
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define CPI_HD __host__ __device__ __forceinline__
#else
#define CPI_HD inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// Scheduling fence: keeps hipcc from hoisting every operand load of a long straight-line block to its top
// (which maximises registers, i.e. minimises co-resident wavefronts, in kernels that are latency-bound).
#ifndef CPI_FENCE_MEM
#define CPI_FENCE_MEM 0   // experiment: the fence also orders MEMORY operations (loads are formed late, where they are used)
#endif
#if CPI_FENCE_MEM
#define CPI_SCHED_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CPI_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#else
#define CPI_SCHED_FENCE() ((void)0)
#endif
// A fence that PINS results: instruction selection orders only memory operations along fences, the arithmetic
// between them floats (hipcc issued the loads of every step of hsn::h1t_vec first -- the whole block table, 216
// registers -- and all the arithmetic after the last fence).  An asm statement that takes a step's results as in / out
// operands must follow their computation and, being volatile with a memory clobber, precedes the next step's loads.
#if defined(__HIP_DEVICE_COMPILE__)
#define CPI_PIN3(a, b, c) do { asm volatile("" : "+v"(a), "+v"(b), "+v"(c) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define CPI_PIN1(a) do { asm volatile("" : "+v"(a) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CPI_PIN3(a, b, c) ((void)0)
#define CPI_PIN1(a) ((void)0)
#endif

namespace cpi {

// ------------------------------------------------------------------------------------------
// small fixed-size types (all register-resident)
struct V3 { double x, y, z; };
struct M3 { double m[3][3]; };  // row-major
struct Q4 { double x, y, z, w; };  // JPL

CPI_HD V3 mk(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
CPI_HD V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
CPI_HD V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
CPI_HD V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
CPI_HD V3 operator*(double s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
CPI_HD V3 axpy(double s, V3 a, V3 b) { return mk(fma(s, a.x, b.x), fma(s, a.y, b.y), fma(s, a.z, b.z)); }
CPI_HD V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
CPI_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
CPI_HD double get(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
CPI_HD V3 unit(int i) { return mk(i == 0 ? 1.0 : 0.0, i == 1 ? 1.0 : 0.0, i == 2 ? 1.0 : 0.0); }

CPI_HD M3 eye() {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = (i == j) ? 1.0 : 0.0;
    return r;
}
CPI_HD M3 zero3() {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = 0.0;
    return r;
}
CPI_HD V3 mul(const M3 &A, V3 v) {  // A v
    return mk(A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z,
              A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
              A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z);
}
CPI_HD V3 mulT(const M3 &A, V3 v) {  // A^T v
    return mk(A.m[0][0] * v.x + A.m[1][0] * v.y + A.m[2][0] * v.z,
              A.m[0][1] * v.x + A.m[1][1] * v.y + A.m[2][1] * v.z,
              A.m[0][2] * v.x + A.m[1][2] * v.y + A.m[2][2] * v.z);
}
CPI_HD M3 mm(const M3 &A, const M3 &B) {  // A B
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            r.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
    return r;
}
CPI_HD M3 mTm(const M3 &A, const M3 &B) {  // A^T B
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            r.m[i][j] = A.m[0][i] * B.m[0][j] + A.m[1][i] * B.m[1][j] + A.m[2][i] * B.m[2][j];
    return r;
}
CPI_HD M3 madd(const M3 &A, const M3 &B) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[i][j] = A.m[i][j] + B.m[i][j];
    return r;
}
CPI_HD V3 col(const M3 &A, int j) { return mk(A.m[0][j], A.m[1][j], A.m[2][j]); }
CPI_HD void addcol(M3 &A, int j, V3 v) { A.m[0][j] += v.x; A.m[1][j] += v.y; A.m[2][j] += v.z; }
// skew_x(w) (quat_ops.h:92-98): skew(w) u = w x u
CPI_HD M3 skew(V3 w) {
    M3 r;
    r.m[0][0] = 0;    r.m[0][1] = -w.z; r.m[0][2] = w.y;
    r.m[1][0] = w.z;  r.m[1][1] = 0;    r.m[1][2] = -w.x;
    r.m[2][0] = -w.y; r.m[2][1] = w.x;  r.m[2][2] = 0;
    return r;
}
// c0*I + c1*[w]x + c2*[w]x^2, with [w]x^2 entries formed exactly as the matrix product w_x*w_x
CPI_HD M3 poly_wx(V3 w, double c0, double c1, double c2) {
    const double xx = w.x * w.x, yy = w.y * w.y, zz = w.z * w.z;
    const double xy = w.x * w.y, xz = w.x * w.z, yz = w.y * w.z;
    M3 r;
    r.m[0][0] = fma(c2, -(yy + zz), c0);
    r.m[1][1] = fma(c2, -(xx + zz), c0);
    r.m[2][2] = fma(c2, -(xx + yy), c0);
    r.m[0][1] = fma(c2, xy, -c1 * w.z); r.m[1][0] = fma(c2, xy, c1 * w.z);
    r.m[0][2] = fma(c2, xz, c1 * w.y);  r.m[2][0] = fma(c2, xz, -c1 * w.y);
    r.m[1][2] = fma(c2, yz, -c1 * w.x); r.m[2][1] = fma(c2, yz, c1 * w.x);
    return r;
}

// ------------------------------------------------------------------------------------------
// sin/cos.  IMU angles |w|*dt are tiny (<= a few 0.1 rad even over dropped samples), so the common
// path is a range-reduction-free Taylor/Horner kernel (truncation < 3e-19 on |x| <= 1).  Larger
// arguments are reduced by a three-term Cody-Waite subtraction of k*pi/2 (exact for |k| < 2^20, i.e.
// |x| < 1.6e6 rad per IMU interval -- far beyond anything physical) onto the same polynomials; there
// is deliberately no Payne-Hanek path: it would double the kernel's register footprint for inputs
// that cannot occur.
// p*z + C as ONE v_fma_f64 on the device.  (Left to itself hipcc turns a Horner step whose addend is a
// loop-invariant constant into v_mov_b64 + v_fmac_f64, i.e. two issue slots per coefficient.)
#if defined(__HIP_DEVICE_COMPILE__)
#define CPI_HORNER(p, z, C)                                                          \
    do {                                                                             \
        double c__ = (C);                                                            \
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "v"(p), "v"(z), "s"(c__));        \
    } while (0)
#else
#define CPI_HORNER(p, z, C) p = fma(p, z, (C))
#endif
// Taylor/Horner kernels.  SHORT (|x| <= 0.25, truncation < 3e-18 relative) is what every physical IMU
// interval needs; LONG covers |x| <= 1.
template <bool LONG>
CPI_HD void sincos_poly(double x, double &s, double &c) {
    const double z = x * x;
    double ps, pc;
    if (LONG) {
        ps = -1.0 / 121645100408832000.0;                 // x^19
        CPI_HORNER(ps, z, 1.0 / 355687428096000.0);      // x^17
        CPI_HORNER(ps, z, -1.0 / 1307674368000.0);       // x^15
        CPI_HORNER(ps, z, 1.0 / 6227020800.0);           // x^13
        CPI_HORNER(ps, z, -1.0 / 39916800.0);            // x^11
        pc = -1.0 / 6402373705728000.0;                   // x^18
        CPI_HORNER(pc, z, 1.0 / 20922789888000.0);       // x^16
        CPI_HORNER(pc, z, -1.0 / 87178291200.0);         // x^14
        CPI_HORNER(pc, z, 1.0 / 479001600.0);            // x^12
    } else {
        ps = 1.0 / 6227020800.0;                          // x^13
        CPI_HORNER(ps, z, -1.0 / 39916800.0);            // x^11
        pc = -1.0 / 87178291200.0;                        // x^14
        CPI_HORNER(pc, z, 1.0 / 479001600.0);            // x^12
    }
    CPI_HORNER(ps, z, 1.0 / 362880.0);                   // x^9
    CPI_HORNER(ps, z, -1.0 / 5040.0);                    // x^7
    CPI_HORNER(ps, z, 1.0 / 120.0);                      // x^5
    CPI_HORNER(ps, z, -1.0 / 6.0);                       // x^3
    s = fma(x * z, ps, x);
    CPI_HORNER(pc, z, -1.0 / 3628800.0);                 // x^10
    CPI_HORNER(pc, z, 1.0 / 40320.0);                    // x^8
    CPI_HORNER(pc, z, -1.0 / 720.0);                     // x^6
    CPI_HORNER(pc, z, 1.0 / 24.0);                       // x^4
    CPI_HORNER(pc, z, -0.5);                             // x^2
    c = fma(z, pc, 1.0);
}
// |x| > 0.25: long polynomial up to 1, Cody-Waite reduction beyond
CPI_HD void sincos_wide(double x, double &s, double &c) {
    if (fabs(x) <= 1.0) { sincos_poly<true>(x, s, c); return; }
    // pi/2 split into three parts with trailing zero bits (the classic fdlibm constants)
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050650619224932e-11, r);
    r = fma(-k, 2.02226624879595063154e-21, r);
    double sr, cr;
    sincos_poly<true>(r, sr, cr);
    const int q = ((int)(long long)k) & 3;
    s = (q == 0) ? sr : ((q == 1) ? cr : ((q == 2) ? -sr : -cr));
    c = (q == 0) ? cr : ((q == 1) ? -sr : ((q == 2) ? -cr : sr));
}
CPI_HD void sincos_fast(double x, double &s, double &c) {
    const bool wide = fabs(x) > 0.25;
#if defined(__HIP_DEVICE_COMPILE__)
    // Every lane of a wave takes the short polynomial for any physical input (|w| dt <= 0.25 rad per interval); the
    // wide path runs under a WAVE-UNIFORM test, so the common case costs one compare and one scalar branch instead of
    // two levels of exec-mask bookkeeping.
    sincos_poly<false>(x, s, c);
    if (__builtin_amdgcn_ballot_w64(wide) != 0) {
        double s2, c2;
        sincos_wide(x, s2, c2);
        s = wide ? s2 : s;
        c = wide ? c2 : c;
    }
#else
    if (wide) sincos_wide(x, s, c); else sincos_poly<false>(x, s, c);
#endif
}
// |w| and 1/|w| from ONE reciprocal-square-root seed + Newton (device); sqrt + divide on the host.
// Host: m2 == 0 (or denormal) returns mag = 0, im = 0.  Device: such an argument is clamped (see below); either way the
// rate is far below the Taylor threshold, whose branch never uses im.
CPI_HD void mag_and_inverse(double m2, double &mag, double &im) {
#if defined(__HIP_DEVICE_COMPILE__)
    // One v_max instead of a compare and six selects: an argument below 1e-280 (exact zero, NaN) is treated as
    // 1e-280, i.e. mag = 1e-140 and im = 1e140 -- every caller either takes its small-angle branch on such a
    // magnitude (|w| < 0.0087) or only ever passes O(1) arguments (rot_2_quat) / checks positivity itself.
    const double a = fmax(m2, 1e-280);
    double y = __builtin_amdgcn_rsq(a);          // ~2^-26 relative
    double g = a * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);          // ~2^-50
    double d = fma(-g, g, a);
    g = fma(d, h, g);                            // sqrt(a), <= 1 ulp
    r = fma(-h, g, 0.5);
    h = fma(h, r, h);                            // 1/(2 sqrt(a)), <= 1 ulp
    mag = g;
    im = 2.0 * h;
#else
    mag = sqrt(m2);
    im = (m2 > 1e-280) ? 1.0 / mag : 0.0;
    if (!(m2 > 1e-280)) mag = 0.0;
#endif
}

// 1 / sqrt(a): the Newton sequence of mag_and_inverse WITHOUT its clamp -- v_rsq_f64 gives NaN for a < 0 and +inf for 0, and
// 0 x inf = NaN carries either through the iteration, so a non-positive or NaN argument poisons whatever is scaled by the result
// (the pivots of cpi_sqrt_info_kernel rely on that; so does the normalisation of a zero quaternion).  <= 1 ulp for a > 1e-280.
#ifndef CPI_QUAT_RECIP
#define CPI_QUAT_RECIP 1
#endif
CPI_HD double inv_sqrt_unclamped(double a) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    const double d = fma(-g, g, a);
    g = fma(d, h, g);
    r = fma(-h, g, 0.5);
    h = fma(h, r, h);
    return 2.0 * h;
#else
    return 1.0 / sqrt(a);
#endif
}

// ------------------------------------------------------------------------------------------
// quaternion helpers (quat_ops.h)
CPI_HD Q4 rot_2_quat(const M3 &R) {  // quat_ops.h:45-86
    // sqrt(x) and 1/sqrt(x) come from ONE v_rsq_f64 seed (mag_and_inverse) instead of a sqrt and a division each:
    // q_c = sqrt(d / 4) = sqrt(d) / 2 and 1 / (4 q_c) = 1 / (2 sqrt(d)); the final normalisation multiplies by 1 / |q|.
    const double r00 = R.m[0][0], r11 = R.m[1][1], r22 = R.m[2][2];
    const double T = r00 + r11 + r22;
    Q4 q;
    double m, im;
    if ((r00 >= T) && (r00 >= r11) && (r00 >= r22)) {
        mag_and_inverse(1 + (2 * r00) - T, m, im);
        q.x = 0.5 * m;
        const double k = 0.5 * im;
        q.y = k * (R.m[0][1] + R.m[1][0]); q.z = k * (R.m[0][2] + R.m[2][0]); q.w = k * (R.m[1][2] - R.m[2][1]);
    } else if ((r11 >= T) && (r11 >= r00) && (r11 >= r22)) {
        mag_and_inverse(1 + (2 * r11) - T, m, im);
        q.y = 0.5 * m;
        const double k = 0.5 * im;
        q.x = k * (R.m[0][1] + R.m[1][0]); q.z = k * (R.m[1][2] + R.m[2][1]); q.w = k * (R.m[2][0] - R.m[0][2]);
    } else if ((r22 >= T) && (r22 >= r00) && (r22 >= r11)) {
        mag_and_inverse(1 + (2 * r22) - T, m, im);
        q.z = 0.5 * m;
        const double k = 0.5 * im;
        q.x = k * (R.m[0][2] + R.m[2][0]); q.y = k * (R.m[1][2] + R.m[2][1]); q.w = k * (R.m[0][1] - R.m[1][0]);
    } else {
        mag_and_inverse(1 + T, m, im);
        q.w = 0.5 * m;
        const double k = 0.5 * im;
        q.x = k * (R.m[1][2] - R.m[2][1]); q.y = k * (R.m[2][0] - R.m[0][2]); q.z = k * (R.m[0][1] - R.m[1][0]);
    }
    if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    double n, in;
    mag_and_inverse(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w, n, in);
    q.x *= in; q.y *= in; q.z *= in; q.w *= in;
    return q;
}
CPI_HD M3 quat_2_Rot(Q4 q) {  // quat_ops.h:104-109
    const V3 v = mk(q.x, q.y, q.z);
    const double c = 2 * q.w * q.w - 1;
    const M3 S = skew(v);
    M3 R;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            R.m[i][j] = c * (i == j ? 1.0 : 0.0) - 2 * q.w * S.m[i][j] + 2 * get(v, i) * get(v, j);
    return R;
}
CPI_HD Q4 quat_multiply(Q4 q, Q4 p) {  // quat_ops.h:115-128
    const V3 qv = mk(q.x, q.y, q.z), pv = mk(p.x, p.y, p.z);
    const V3 c = cross(qv, pv);
    Q4 r;
    r.x = q.w * p.x - c.x + q.x * p.w;
    r.y = q.w * p.y - c.y + q.y * p.w;
    r.z = q.w * p.z - c.z + q.z * p.w;
    r.w = q.w * p.w - dot(qv, pv);
    if (r.w < 0) { r.x = -r.x; r.y = -r.y; r.z = -r.z; r.w = -r.w; }
    const double n2 = r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w;
#if defined(__HIP_DEVICE_COMPILE__) && CPI_QUAT_RECIP
    // q / |q| as q * (1 / |q|), 1 / |q| from one v_rsq_f64 seed (<= 1 ulp; a zero product still normalises to NaN as the reference's
    // 0 / 0 does): 15 instructions where an IEEE square root and four IEEE divisions took ~60, five times per factor -- the sweeps that
    // are bound by their instruction count (Hessian: 1 754 -> see profiles/r06_small_sweeps.md) feel it; <= 2 ulp per component.
    const double in = inv_sqrt_unclamped(n2);
    r.x *= in; r.y *= in; r.z *= in; r.w *= in;
#else
    const double n = sqrt(n2);
    r.x /= n; r.y /= n; r.z /= n; r.w /= n;
#endif
    return r;
}
CPI_HD Q4 quat_inv(Q4 q) { Q4 r; r.x = -q.x; r.y = -q.y; r.z = -q.z; r.w = q.w; return r; }
CPI_HD M3 Exp_so3(V3 w) {  // quat_ops.h:145-162
    double s, c;
#if defined(__HIP_DEVICE_COMPILE__) && CPI_QUAT_RECIP
    // |w| and 1 / |w| from one seed (mag_and_inverse: w = 0 is clamped to |w| = 1e-140, whose coefficients sin(t) / t = 1 and
    // (1 - cos t) / t^2 = 0 give the identity as quat_ops.h:148-150 does) instead of a square root and two divisions
    double theta, ith;
    mag_and_inverse(dot(w, w), theta, ith);
    sincos_fast(theta, s, c);
    return poly_wx(w, 1.0, s * ith, ((1 - c) * ith) * ith);
#else
    const double theta = sqrt(dot(w, w));
    // theta == 0 returns the identity (quat_ops.h:148-150); branch-free: with w = 0 any finite coefficients give I
    const double th = (theta == 0) ? 1.0 : theta;
    sincos_fast(theta, s, c);
    return poly_wx(w, 1.0, s / th, (1 - c) / (th * th));
#endif
}

// ------------------------------------------------------------------------------------------
// Per-interval closed-form coefficients (CpiV1.h:97-142, identical in CpiV2.h:118-176).
struct StepCoef {
    double dt, mag, im, wdt, sn, cs;
    double s1, s2;          // R_tau2tau1 = I - s1 [w]x + s2 [w]x^2
    double f1, f2, f3, f4;  // alpha_arg = dt^2/2 I + f1 [w]x + f2 [w]x^2 ; Beta_arg = dt I + f3 [w]x + f4 [w]x^2
    bool small;
};
CPI_HD StepCoef step_coef(V3 w, double dt) {
    StepCoef k;
    k.dt = dt;
    mag_and_inverse(dot(w, w), k.mag, k.im);
    const double im = k.im;
    k.wdt = k.mag * dt;
    k.small = (k.mag < 0.008726646);  // threshold on the rate, CpiV1.h:101
    sincos_fast(k.wdt, k.sn, k.cs);
    // Closed forms first (CpiV1.h:120,137-141); the Taylor forms of the |w| < 0.0087 rad/s branch (:119,132-135)
    // are patched in under a WAVE-UNIFORM test, so a wavefront without such a lane -- every wavefront of a real,
    // noisy IMU stream -- skips them.  (1/3, 1/6 as constants: <= 1 ulp from the reference's divisions.)
    const double im2 = im * im, im3 = im2 * im, im4 = im2 * im2;
    const double omc = 1.0 - k.cs;
    k.s1 = k.sn * im;
    k.s2 = omc * im2;
    k.f1 = (k.wdt * k.cs - k.sn) * im3;
    k.f2 = (k.wdt * k.wdt - 2 * k.cs - 2 * k.wdt * k.sn + 2) * (0.5 * im4);
    k.f3 = -omc * im2;
    k.f4 = (k.wdt - k.sn) * im3;
    const bool sm = k.small;
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_amdgcn_ballot_w64(sm) != 0)
#else
    if (sm)
#endif
    {
        const double t2 = dt * dt, t3 = t2 * dt;
        k.s1 = sm ? dt : k.s1;
        k.s2 = sm ? 0.5 * t2 : k.s2;
        k.f1 = sm ? -(t3 * (1.0 / 3.0)) : k.f1;
        k.f2 = sm ? (t2 * t2 * 0.125) : k.f2;
        k.f3 = sm ? -(0.5 * t2) : k.f3;
        k.f4 = sm ? (t3 * (1.0 / 6.0)) : k.f4;
    }
    return k;
}
// (Tried and rejected: power series in (|w| dt)^2 for s1, s2, f1..f4 on the mean-only path -- no square root, no
// reciprocal, no sin / cos, ~20 fewer FP64 instructions per interval.  They are MORE accurate than the reference's
// closed forms, and that is the problem: just above its 0.0087 rad/s Taylor threshold the reference's f2 =
// (x^2 - 2 cos x - 2 x sin x + 2) / (2 |w|^4) is cancellation noise of order 1e-16 / |w|^4, which the closed forms
// above reproduce (same expressions, parity 2e-15) and a series does not: alpha moved by 6e-11 on the golden windows
// with |w| ~ 0.01 rad/s and would pass the 1e-9 gate's margin with larger specific forces or longer windows.
// Parity with the reference, noise included, comes first.)
CPI_HD M3 R_step_of(V3 w, const StepCoef &k) { return poly_wx(w, 1.0, -k.s1, k.s2); }
// rotation over the first half of the interval (CpiV1.h:267-268 / CpiV2.h:315-322)
CPI_HD M3 R_half_of(V3 w, const StepCoef &k) {
    const double h = 0.5 * k.dt;
    if (k.small) return poly_wx(w, 1.0, -h, h * h / 2);
    double s, c;
    sincos_fast(k.mag * h, s, c);
    const double im = k.im;
    return poly_wx(w, 1.0, -s * im, (1.0 - c) * im * im);
}
// ua = alpha_arg * a, ub = Beta_arg * a in vector form ([w]x^2 a = w x (w x a))
CPI_HD void arg_times(V3 w, V3 a, const StepCoef &k, V3 &ua, V3 &ub) {
    const V3 wa = cross(w, a), wwa = cross(w, wa);
    ua = axpy(k.f2, wwa, axpy(k.f1, wa, (0.5 * k.dt * k.dt) * a));
    ub = axpy(k.f4, wwa, axpy(k.f3, wa, k.dt * a));
}

// ------------------------------------------------------------------------------------------
// Sequential mean (+ analytic Jacobian) recursion of one lane (kernel "cpi_mean").
template <bool JAC>
struct MeanState {
    M3 R;
    V3 alpha, beta;
    double DT;
    M3 Jq, Ja, Jb, Ha, Hb, Oa, Ob;  // only touched when JAC
};
template <bool JAC>
CPI_HD void mean_init(MeanState<JAC> &s) {
    s.R = eye();
    s.alpha = mk(0, 0, 0); s.beta = mk(0, 0, 0); s.DT = 0;
    if (JAC) { s.Jq = zero3(); s.Ja = zero3(); s.Jb = zero3(); s.Ha = zero3(); s.Hb = zero3(); s.Oa = zero3(); s.Ob = zero3(); }
}

// One feed_IMU.  MODEL 1: CpiV1.h:62-154(+161-259 if JAC).  MODEL 2: CpiV2.h:84-186(+187-305 if JAC,
// i.e. the analytic Jacobians incl. O_a/O_b used when state_transition_jacobians == false).
// w0/a0/w1/a1 are RAW readings; bw/ba the linearisation biases; gk = R(q_k_lin)*grav (model 2).
template <int MODEL, bool JAC, bool AVG>
CPI_HD void mean_step(MeanState<JAC> &s, double t0, double t1, V3 w0, V3 a0, V3 w1, V3 a1, V3 bw, V3 ba, V3 gk,
                      bool active = true) {
    double dt = t1 - t0;
    // dt == 0: feed_IMU returns early (CpiV1.h:72); dt < 0: caller skips (GraphSolver_IMU.cpp:52); NaN: separator.
    // Mean-only path: a dt == 0 step is an exact no-op of the arithmetic below (every increment is a product
    // with 0), so inactive / skipped intervals run branch-free with dt forced to 0 -- no divergence, no
    // exec-mask bookkeeping, no register copies at control-flow joins.
    if (!JAC) { if (!(active && dt > 0)) dt = 0; }
    else if (!(active && dt > 0)) return;
    s.DT += dt;
    V3 w = w0 - bw;
    V3 a = a0 - ba;
    V3 gtau = mk(0, 0, 0);
    if (MODEL == 2) { gtau = mul(s.R, gk); a = a - gtau; }
    if (AVG) {
        w = 0.5 * (w + (w1 - bw));
        if (MODEL == 1) a = 0.5 * (a + (a1 - ba));
    }
    const StepCoef k = step_coef(w, dt);
    if (!JAC) {
        // Mean-only: rotate the columns of R in place, R'[:,c] = r - s1 (w x r) + s2 (w x (w x r)),
        // instead of forming R_tau2tau1 and a 3x3 product -- same arithmetic, half the live registers.
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const V3 r = col(s.R, c);
            const V3 wr = cross(w, r), wwr = cross(w, wr);
            const V3 rn = axpy(k.s2, wwr, axpy(-k.s1, wr, r));
            s.R.m[0][c] = rn.x; s.R.m[1][c] = rn.y; s.R.m[2][c] = rn.z;
        }
        if (MODEL == 2 && AVG) a = 0.5 * (a + (a1 - ba - mul(s.R, gk)));
        V3 ua, ub;
        arg_times(w, a, k, ua, ub);
        s.alpha = s.alpha + (dt * s.beta + mulT(s.R, ua));  // uses the not-yet-updated beta (CpiV1.h:153)
        s.beta = s.beta + mulT(s.R, ub);
        return;
    }
    const M3 Rs = R_step_of(w, k);
    const M3 Rn = mm(Rs, s.R);
    if (MODEL == 2 && AVG) a = 0.5 * (a + (a1 - ba - mul(Rn, gk)));
    V3 ua, ub;
    arg_times(w, a, k, ua, ub);
    const V3 da = mulT(Rn, ua), db = mulT(Rn, ub);
    s.alpha = s.alpha + (dt * s.beta + da);  // uses the not-yet-updated beta (CpiV1.h:153)
    s.beta = s.beta + db;

    if (JAC) {
        // right Jacobian times dt (CpiV1.h:162-167)
        double c1, c2;
        if (k.small) { c1 = 0.5; c2 = 1.0 / 6.0; }
        else { const double iw = 1.0 / k.wdt; c1 = (1 - k.cs) * iw * iw; c2 = (k.wdt - k.sn) * iw * iw * iw; }
        const M3 Jr_dt = poly_wx(w, dt, -c1 * dt * dt, c2 * dt * dt * dt);
        const M3 Jsave = s.Jq;
        s.Jq = madd(mm(Rs, s.Jq), Jr_dt);
        // accel-bias Jacobians (CpiV1.h:170-172)
        const M3 alpha_arg = poly_wx(w, 0.5 * dt * dt, k.f1, k.f2);
        const M3 Beta_arg = poly_wx(w, dt, k.f3, k.f4);
        const M3 Hal = mTm(Rn, alpha_arg), Hbe = mTm(Rn, Beta_arg);
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
            s.Ha.m[i][j] = (s.Ha.m[i][j] - Hal.m[i][j]) + dt * s.Hb.m[i][j];
            s.Hb.m[i][j] -= Hbe.m[i][j];
        }
        if (MODEL == 2) {  // CpiV2.h:201-205
            const M3 Sg = skew(gk);
            const M3 RS = mm(s.R, Sg);
            const M3 Ta = mm(Hal, RS), Tb = mm(Hbe, RS);
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                s.Oa.m[i][j] = (s.Oa.m[i][j] + dt * s.Ob.m[i][j]) - Ta.m[i][j];
                s.Ob.m[i][j] -= Tb.m[i][j];
            }
        }
        // d f_k / d|w| (CpiV1.h:196-238)
        double d1, d2, d3, d4;
        if (k.small) {
            const double t2 = dt * dt, t4 = t2 * t2;
            d1 = -(t4 * dt / 15); d2 = (t4 * t2 / 72); d3 = -(t4 / 12); d4 = (t4 * dt / 60);
        } else {
            const double im = k.im, im2 = im * im, im4 = im2 * im2, im5 = im4 * im, im6 = im4 * im2;
            const double x = k.wdt, x2 = x * x;
            d1 = (x2 * k.sn - 3 * k.sn + 3 * x * k.cs) * im5;
            d2 = (x2 - 4 * k.cs - 4 * x * k.sn + x2 * k.cs + 4) * im6;
            d3 = (2 * (k.cs - 1) + x * k.sn) * im4;
            d4 = (2 * x + x * k.cs - 3 * k.sn) * im5;
        }
        // gyro-bias Jacobians, column by column (CpiV1.h:241-259):
        //   (d_R_bw_i*arg + R^T*G_i) a = R^T ( G_i a - (J_q e_i) x (arg a) )
        const V3 wa = cross(w, a), wwa = cross(w, wa);
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) s.Ja.m[i][j] += s.Jb.m[i][j] * dt;  // old J_b
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const V3 e = unit(c);
            const double wc = get(w, c);
            const V3 ea = cross(e, a);
            const V3 sym = cross(e, wa) + cross(w, ea);  // (e_x w_x + w_x e_x) a
            const V3 Ga = axpy(wc * d1, wa, axpy(wc * d2, wwa, -(k.f1 * ea) - (k.f2 * sym)));
            const V3 Gb = axpy(wc * d3, wa, axpy(wc * d4, wwa, -(k.f3 * ea) - (k.f4 * sym)));
            const V3 jq = col(s.Jq, c);  // updated J_q (CpiV1.h:175-177)
            V3 ca = mulT(Rn, Ga - cross(jq, ua));
            V3 cb = mulT(Rn, Gb - cross(jq, ub));
            if (MODEL == 2) {  // CpiV2.h:282-305 (the J_b column-0 term carries the reference's double minus)
                const V3 t = cross(col(Jsave, c), gtau);
                const V3 ga = mul(Hal, t), gb = mul(Hbe, t);
                ca = ca - ga;
                cb = (c == 0) ? cb + gb : cb - gb;
            }
            addcol(s.Ja, c, ca);
            addcol(s.Jb, c, cb);
        }
    }
    s.R = Rn;
}

// Order-preserving composition of two consecutive segments A (earlier) then B (later), each
// integrated from identity/zero (model 1): R_AB = R_B R_A, beta = beta_A + R_A^T beta_B,
// alpha = alpha_A + beta_A DT_B + R_A^T alpha_B, and for the analytic Jacobians
//   J_q = R_B J_q^A + J_q^B,  H_b = H_b^A + R_A^T H_b^B,  H_a = H_a^A + H_b^A DT_B + R_A^T H_a^B,
//   J_b = J_b^A + R_A^T (J_b^B + [beta_B]x J_q^A),  J_a = J_a^A + J_b^A DT_B + R_A^T (J_a^B + [alpha_B]x J_q^A).
template <bool JAC>
CPI_HD void mean_combine(MeanState<JAC> &A, const MeanState<JAC> &B) {
    if (JAC) {
        const M3 SbJ = mm(skew(B.beta), A.Jq), SaJ = mm(skew(B.alpha), A.Jq);
        const M3 nJb = mTm(A.R, madd(B.Jb, SbJ));
        const M3 nJa = mTm(A.R, madd(B.Ja, SaJ));
        const M3 nHb = mTm(A.R, B.Hb), nHa = mTm(A.R, B.Ha);
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
            A.Ja.m[i][j] = (A.Ja.m[i][j] + A.Jb.m[i][j] * B.DT) + nJa.m[i][j];
            A.Jb.m[i][j] += nJb.m[i][j];
            A.Ha.m[i][j] = (A.Ha.m[i][j] + A.Hb.m[i][j] * B.DT) + nHa.m[i][j];
            A.Hb.m[i][j] += nHb.m[i][j];
        }
        A.Jq = madd(mm(B.R, A.Jq), B.Jq);
    }
    A.alpha = A.alpha + (B.DT * A.beta + mulT(A.R, B.alpha));
    A.beta = A.beta + mulT(A.R, B.beta);
    A.R = mm(B.R, A.R);
    A.DT += B.DT;
}

// ------------------------------------------------------------------------------------------
// Model-2 means over a SEGMENT of a window (several lanes per window).  Model 2 integrates the local specific force
// a_hat = a_m - b_a - R g_k (CpiV2.h:99), which depends on the rotation R accumulated since the START OF THE WINDOW --
// unknown to a lane that starts in the middle.  It enters linearly: with R = R_loc R_A (R_A = everything before the
// segment) and g' = R_A g_k,
//     beta_seg(g')  = beta0  - Gam g',     Gam = sum_i R_new,i^T Beta_arg,i  Rg_i
//     alpha_seg(g') = alpha0 - Lam g',     Lam: same recursion as alpha with (Gam, alpha_arg) in place of (beta, .)
// where beta0 / alpha0 are integrated from a_m - b_a alone, all rotations are the segment's own, and
// Rg_i = R_old,i (or (R_old,i + R_new,i)/2 with imu_avg, CpiV2.h:146-149).  Segments compose like the means,
//     Gam_AB = Gam_A + R_A^T Gam_B R_A,      Lam_AB = Lam_A + Gam_A DT_B + R_A^T Lam_B R_A,
// and the window's means are beta0 - Gam g_k, alpha0 - Lam g_k.
struct GravAcc { M3 Gam, Lam; };
CPI_HD void grav_init(GravAcc &g) { g.Gam = zero3(); g.Lam = zero3(); }
// One interval of a model-2 segment: means from the raw specific force into s (R local), gravity response into g.
template <bool AVG>
CPI_HD void mean_step_v2seg(MeanState<false> &s, GravAcc &g, double t0, double t1, V3 w0, V3 a0, V3 w1, V3 a1,
                            V3 bw, V3 ba, bool active = true) {
    double dt = t1 - t0;
    if (!(active && dt > 0)) dt = 0;   // exact no-op, as in mean_step
    s.DT += dt;
    V3 w = w0 - bw;
    V3 a = a0 - ba;
    if (AVG) { w = 0.5 * (w + (w1 - bw)); a = 0.5 * (a + (a1 - ba)); }
    const StepCoef k = step_coef(w, dt);
    M3 Rg = s.R;                       // R_old (local)
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const V3 r = col(s.R, c);
        const V3 wr = cross(w, r), wwr = cross(w, wr);
        const V3 rn = axpy(k.s2, wwr, axpy(-k.s1, wr, r));
        s.R.m[0][c] = rn.x; s.R.m[1][c] = rn.y; s.R.m[2][c] = rn.z;
    }
    if (AVG) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) Rg.m[i][j] = 0.5 * (Rg.m[i][j] + s.R.m[i][j]);
    }
    V3 ua, ub;
    arg_times(w, a, k, ua, ub);
    s.alpha = s.alpha + (dt * s.beta + mulT(s.R, ua));
    s.beta = s.beta + mulT(s.R, ub);
    // gravity response: columns of alpha_arg Rg and Beta_arg Rg, then R_new^T (.)
    M3 Am, Bm;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        V3 ca, cb;
        arg_times(w, col(Rg, c), k, ca, cb);
        Am.m[0][c] = ca.x; Am.m[1][c] = ca.y; Am.m[2][c] = ca.z;
        Bm.m[0][c] = cb.x; Bm.m[1][c] = cb.y; Bm.m[2][c] = cb.z;
    }
    const M3 dL = mTm(s.R, Am), dG = mTm(s.R, Bm);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            g.Lam.m[i][j] = (g.Lam.m[i][j] + dt * g.Gam.m[i][j]) + dL.m[i][j];   // old Gam, like alpha uses old beta
            g.Gam.m[i][j] += dG.m[i][j];
        }
}
// A (earlier) o B (later); call BEFORE mean_combine(A, B) (needs A.R and B.DT as they are).
CPI_HD void grav_combine(GravAcc &gA, const MeanState<false> &A, const GravAcc &gB, const MeanState<false> &B) {
    const M3 tG = mm(mTm(A.R, gB.Gam), A.R), tL = mm(mTm(A.R, gB.Lam), A.R);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            gA.Lam.m[i][j] = (gA.Lam.m[i][j] + gA.Gam.m[i][j] * B.DT) + tL.m[i][j];
            gA.Gam.m[i][j] += tG.m[i][j];
        }
}
// Window result from the composed segment: subtract the gravity response to g_k = R(q_k_lin) g.
CPI_HD void grav_apply(MeanState<false> &s, const GravAcc &g, V3 gk) {
    s.beta = s.beta - mul(g.Gam, gk);
    s.alpha = s.alpha - mul(g.Lam, gk);
}

// ------------------------------------------------------------------------------------------
// Covariance / state-transition column recursion (kernel "cpi_cov").
//
// One lane owns ONE COLUMN x of the (symmetric) covariance -- or, for model 2, one column of the
// compounded state-transition matrix Discrete_J_b -- as a vector over the error state
//   [theta(0:3) b_w(3:6) v(6:9) b_a(9:12) p(12:15) | theta_clone(15:18) | theta_klin(18:21)].
// With F's block structure (CpiV1.h:276-281, CpiV2.h:331-338) the product F x is lane-local:
//   (F x)_theta = -w x x_theta - x_bw
//   (F x)_v     = -Rs^T ( a x x_theta + x_ba + g_tau x x_clone + R_old (g_k x x_klin) )
//   (F x)_p     = x_v                                  (all other rows are zero)
// and P F^T is its transpose, which the kernel obtains through a 9-row LDS exchange.
// Classic RK4 with stage rotations (R_old, R_mid, R_mid, R_new), exactly as the reference.

// Per-sample broadcast record produced lane-parallel by phase A of the kernel.
struct SampleRec {
    double dt;
    V3 w, a0, a1;       // w_hat; a0 = a_m_0 - b_a (model-1 averaging already applied); a1 = a_m_1 - b_a (model 2 + avg)
    double f1, f2, f3, f4;
    M3 Rstep, Rhalf;
};
static const int SAMPLE_REC_DOUBLES = 1 + 9 + 4 + 18;  // 32

template <int MODEL, bool AVG>
CPI_HD SampleRec make_sample_rec(double t0, double t1, V3 w0, V3 a0, V3 w1, V3 a1, V3 bw, V3 ba) {
    SampleRec r;
    double dt = t1 - t0;
    if (!(dt > 0)) dt = 0;  // a dt == 0 step is an exact no-op of the recursion below
    r.dt = dt;
    V3 w = w0 - bw;
    r.a0 = a0 - ba;
    r.a1 = a1 - ba;
    if (AVG) {
        w = 0.5 * (w + (w1 - bw));
        if (MODEL == 1) r.a0 = 0.5 * (r.a0 + r.a1);
    }
    r.w = w;
    const StepCoef k = step_coef(w, dt);
    r.f1 = k.f1; r.f2 = k.f2; r.f3 = k.f3; r.f4 = k.f4;
    r.Rstep = R_step_of(w, k);
    r.Rhalf = R_half_of(w, k);
    return r;
}

template <int MODEL>
struct CovDims {
    static const int NR = (MODEL == 1) ? 15 : 18;    // dynamic rows carried per column
    static const int NPCOL = NR;                     // covariance columns (lanes)
    static const int NDCOL = (MODEL == 1) ? 0 : 9;   // state-transition columns: b_w(3), b_a(3), theta_klin(3)
    static const int NCOL = NPCOL + NDCOL;
    static const int GROUP = (MODEL == 1) ? 16 : 32; // lanes per window
};

CPI_HD M3 rec_mat(const double *rp, int at) {
    M3 A;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) A.m[i][k] = rp[at + i * 3 + k];
    return A;
}
CPI_HD void rec_put_mat(double *rp, int at, const M3 &A) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) rp[at + i * 3 + k] = A.m[i][k];
}
CPI_HD V3 rec_v3(const double *rp, int at) { return mk(rp[at], rp[at + 1], rp[at + 2]); }
CPI_HD void put3(double *o, V3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

// ---- phase A -> phase C interval record (doubles, tightly packed: the reads are 8-byte-aligned ds_read2_b64 pairs).
// Phase A does ALL the work that the lanes of a group share -- closed forms, the running rotation (a prefix
// product over the chunk's intervals), the three RK4 stage rotations, gravity terms and the mean increments --
// once per interval, lane-parallel over intervals; phase C only reads it back as LDS broadcasts.  A small record
// matters twice: fewer LDS bytes per interval, and more intervals per phase-A pass in the same LDS budget.
template <int MODEL> struct IrL;
// R_old is not stored: it is the previous interval's R_new (phase C carries it in registers), and for the first
// interval of a pass the rotation carried in from the previous pass (GS_R0 of the group's carry record).
template <> struct IrL<1> {   // dt, w, a, R_mid, R_new
    static const int DT = 0, W = 1, A = 4, RMID = 7, RNEW = 16, SIZE = 25;
    static const int GTAU = 0, H = 0, ZERO = 0;   // unused
};
template <> struct IrL<2> {   // + g_tau, h_l = R_old (g_k x e_l) for l = 0..2, three zeros
    static const int DT = 0, W = 1, A = 4, GTAU = 7, RMID = 10, RNEW = 19, H = 28, ZERO = 37, SIZE = 40;
};
template <int MODEL> struct IrSize { static const int V = IrL<MODEL>::SIZE; };   // doubles per record
// LDS pitch of a record.  Both pitches keep every record on a 16-B boundary so that phase C's broadcast reads are
// ds_read_b128 (4 LDS cycles per two doubles; the ds_read2_b64 pairs an 8-B aligned record is read with cost 8), and
// both walk the lanes of a phase-A ds_write_b128 group (8 lanes on 32 banks) over distinct 16-B slots: model 1 26
// doubles (25 used; 1.436 -> 1.410 ms per 100 k x 50 against the 25-double pitch), model 2 42 (40 used; 40 = 320 B
// puts every other lane on the same banks).
template <int MODEL> struct IrPitch { static const int V = (MODEL == 1) ? 26 : 42; };
// group-shared carry across chunks: running rotation and means
static const int GS_R = 0, GS_ALPHA = 10, GS_BETA = 14, GS_DT = 18, GS_GK = 20 /* model 2: R(q_k_lin) g */,
                 GS_R0 = 24 /* rotation at the start of the current phase-A pass */, GS_DOUBLES = 34;
CPI_HD void cov_gs_init(double *gs) {
#pragma unroll
    for (int i = 0; i < GS_DOUBLES; i++) gs[i] = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
}
// The mean increment of one interval in the window-start frame, as a composable element:
// (beta, alpha, dt) o (beta', alpha', dt') = (beta + beta', alpha + dt' beta + alpha', dt + dt')   (CpiV1.h:153-154)
struct MeanInc { V3 beta, alpha; double dt; };
CPI_HD MeanInc inc_combine(const MeanInc &A, const MeanInc &B) {   // A earlier, B later
    MeanInc r;
    r.alpha = A.alpha + (B.dt * A.beta + B.alpha);
    r.beta = A.beta + B.beta;
    r.dt = A.dt + B.dt;
    return r;
}
// Finish one interval once its start rotation R_old is known: writes the phase-C record and returns the
// interval's mean increment.  (CpiV1.h:123-154,267-269 / CpiV2.h:99,141-186,315-323)
template <int MODEL, bool AVG>
CPI_HD MeanInc finish_interval(const SampleRec &r, const M3 &R_old, V3 gk, double *ir) {
    const M3 R_new = mm(r.Rstep, R_old), R_mid = mm(r.Rhalf, R_old);
    V3 a = r.a0, gtau = mk(0, 0, 0);
    if (MODEL == 2) {
        gtau = mul(R_old, gk);
        a = a - gtau;
        if (AVG) a = 0.5 * (a + (r.a1 - mul(R_new, gk)));   // CpiV2.h:146-149: average the LOCAL acceleration
    }
    StepCoef k;
    k.dt = r.dt; k.f1 = r.f1; k.f2 = r.f2; k.f3 = r.f3; k.f4 = r.f4;
    V3 ua, ub;
    arg_times(r.w, a, k, ua, ub);
    MeanInc inc;
    inc.alpha = mulT(R_new, ua);
    inc.beta = mulT(R_new, ub);
    inc.dt = r.dt;
    typedef IrL<MODEL> IR;
    ir[IR::DT] = r.dt;
    put3(ir + IR::W, r.w);
    put3(ir + IR::A, a);
    rec_put_mat(ir, IR::RMID, R_mid);
    rec_put_mat(ir, IR::RNEW, R_new);
    if (MODEL == 2) {
        put3(ir + IR::GTAU, gtau);
#pragma unroll
        for (int l = 0; l < 3; l++) put3(ir + IR::H + 3 * l, mul(R_old, cross(gk, unit(l))));
        ir[IR::ZERO] = 0.0; ir[IR::ZERO + 1] = 0.0; ir[IR::ZERO + 2] = 0.0;
    }
    return inc;
}
// Fold a chunk's composed increment into the carried means (gs): alpha += dt_c beta + alpha_c, beta += beta_c.
CPI_HD void gs_apply_inc(double *gs, const MeanInc &c) {
    const V3 beta = rec_v3(gs, GS_BETA);
    put3(gs + GS_ALPHA, rec_v3(gs, GS_ALPHA) + (c.dt * beta + c.alpha));
    put3(gs + GS_BETA, beta + c.beta);
    gs[GS_DT] += c.dt;
}

// Exchange-buffer rows: per window group rows 0..8 = rows (theta,v,p) of F X, rewritten every stage; then,
// shared by all groups of the wavefront, rows 9..14 = constant rows q_j e_j for the b_w / b_a covariance columns
// (their only k contribution is the process noise on the diagonal) and row 15 = zeros (columns with no
// transposed row: clone and transition lanes).
static const int EXCH_GROUP_ROWS = 9, EXCH_SHARED_ROWS = 7;
static const int EXCH_PITCH = 18;  // doubles: 16-B aligned rows, 9 LDS slots (16 B each) apart -- 9 is coprime to 16
// Placement against LDS bank conflicts (64 x 4-B banks = 16 slots of 16 B; a ds_read_b128 is served in four groups of
// 16 lanes, {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32 -- MI355X_MICROARCH.md "LDS"): a lane reads 8
// consecutive slots of "its" row, so a lane group is conflict-free when its 16 row bases are distinct mod 16 slots (or
// identical).  With rows 9 slots apart, the theta/v/p rows of a window sit on residues w + {0,9,2,11,4,13,6,15,8} and
// the shared rows on s + {0,9,2,11,4,13,6}; the lane groups above mix exactly such sets, and they are disjoint for
// every group iff all windows start on the same residue (window stride = 0 mod 256 B) and s = w + 1.  Hence:
//   window g at doubles g*192 (1536 B: 9 rows + a 30-double gap); shared row 0 inside window 0's gap (offset 162);
//   shared rows 1..6 contiguous after the windows at +20 doubles (slot 10 = 1 + 9).
// Measured (rocprofv3 SQ_LDS_BANK_CONFLICT, 100 k windows x 50): see DESIGN.md 3.2.
static const int EXCH_WIN = 192;
// Position of column c inside an exchange row.  Model 2 spreads its 18 columns over lanes 0-2 | 4-6 (clone columns
// 15-17) | 8-19 (columns 3-14); a ds_write_b64 is served 16 contiguous lanes at a time on 32 banks, so the columns the
// first 16 lanes own (0-10 and 15-17) must sit on distinct positions mod 16: the clone columns move to 11-13 and
// columns 11-14 to 14-17.  (With the identity, columns 16 and 17 collide with 0 and 1 on every exchange write.)
#ifndef CPI_COV2_PSYM
#define CPI_COV2_PSYM 0   // model 2: rows p of F X by symmetry + DPP (needs the lane map that puts v and p in one DPP row); see CovPBySymmetry
#endif
#if CPI_COV2_PSYM
// lane map with v / p in DPP row 1 (cov_col_of_lane): the first 16 lanes own columns 0-5, 9-11 and the clone columns 15-17,
// the second 16 own 6-8 and 12-14: clone columns at positions 12-14, p columns at 15-17 keep both groups distinct mod 16
template <int MODEL>
CPI_HD constexpr int exch_pos(int c) { return (MODEL == 1 || c <= 11) ? c : (c <= 14 ? c + 3 : c - 3); }
#else
template <int MODEL>
CPI_HD constexpr int exch_pos(int c) { return (MODEL == 1 || c <= 10) ? c : (c <= 14 ? c + 3 : c - 4); }
#endif
CPI_HD constexpr int exch_doubles(int G) { return G * EXCH_WIN + 20 + (EXCH_SHARED_ROWS - 1) * EXCH_PITCH; }
CPI_HD int exch_shared_off(int G, int s) {   // offset (doubles) of shared row s = layout row 9 + s
    return (s == 0) ? EXCH_GROUP_ROWS * EXCH_PITCH : G * EXCH_WIN + 20 + (s - 1) * EXCH_PITCH;
}

template <int MODEL>
struct CovLane {
    double P0[CovDims<MODEL>::NR];   // column at the start of the interval
    double X[CovDims<MODEL>::NR];    // RK4 stage value
    double acc[CovDims<MODEL>::NR];  // running RK4 sum
    V3 hqt, hqv;                     // half the process-noise variance on this column's own diagonal row
                                     // (theta / v columns only): k_jj = (M_jj + q/2) + (M_jj + q/2)
    // per-interval scratch (read from the interval record)
    V3 w, a, gtau, h;
    double dt;
};

// Column owned by lane l of a window group.  Model 1: lane = column (lane 15 idle).  Model 2: the three
// clone columns (15,16,17) sit exactly one 4-lane DPP bank above the theta columns they are copied from,
//   lanes 0-2 theta | 3 idle | 4-6 clone | 7 idle | 8-19 columns 3..14 | 20-28 transition columns | 29-31 idle
// so the per-interval column clone (CpiV2.h:436-441) is a single masked row_shr:4 DPP move per register instead of a
// round trip through the LDS crossbar.  Idle lanes get column index NCOL.
template <int MODEL>
CPI_HD int cov_col_of_lane(int l) {
    typedef CovDims<MODEL> D;
    if (MODEL == 1) return (l < D::NCOL) ? l : D::NCOL;
#if CPI_COV2_PSYM
    // DPP row 0: theta 0-2 | idle | clone 4-6 | idle | b_w 8-10 | b_a 11-13 | idle 14, 15
    // DPP row 1: transition 16-21 | v 22-24 | transition 25-27 | p 28-30 | idle 31   -- p lanes = v lanes + 6 inside one row
    if (l < 3) return l;
    if (l == 3 || l == 7 || l == 14 || l == 15 || l == 31) return D::NCOL;
    if (l < 7) return 15 + (l - 4);
    if (l < 11) return 3 + (l - 8);
    if (l < 14) return 9 + (l - 11);
    if (l < 22) return 18 + (l - 16);
    if (l < 25) return 6 + (l - 22);
    if (l < 28) return 24 + (l - 25);
    return 12 + (l - 28);
#else
    if (l < 3) return l;
    if (l == 3 || l == 7 || l >= 29) return D::NCOL;
    if (l < 7) return 15 + (l - 4);
    if (l < 20) return l - 5;
    return 18 + (l - 20);
#endif
}
// Which exchange row this column reads as its transposed contribution.
CPI_HD int cov_exch_row(int j) {
    return (j < 3) ? j : ((j >= 6 && j < 9) ? j - 3 : ((j >= 12 && j < 15) ? j - 6 : -1));
}
template <int MODEL>
CPI_HD int cov_read_row(int j) {
    typedef CovDims<MODEL> D;
    if (j >= D::NPCOL) return 15;
    const int er = cov_exch_row(j);
    if (er >= 0) return er;
    if (j >= 3 && j < 6) return 9 + (j - 3);
    if (j >= 9 && j < 12) return 12 + (j - 9);
    return 15;
}
// Offset (doubles, inside an interval record) of this column's h = R_old (g_k x x_klin): the theta_klin
// transition columns (model 2, columns NPCOL+6..8) own one of the three stored vectors, everyone else zero.
template <int MODEL>
CPI_HD int cov_h_offset(int j) {
    typedef CovDims<MODEL> D;
    const int d = j - D::NPCOL;
    return (MODEL == 2 && d >= 6 && d < 9) ? IrL<MODEL>::H + 3 * (d - 6) : IrL<MODEL>::ZERO;
}
// Initialise the shared constant rows (9..15).  Called by every lane for its own column index (all groups write
// identical values); the rows must have been zeroed before.  G = windows per wavefront (layout above).
template <int MODEL>
CPI_HD void cov_exch_init(double *exch, int G, int j, const double q4[4]) {
    if (j >= 3 && j < 6) exch[exch_shared_off(G, j - 3) + exch_pos<MODEL>(j)] = q4[1];
    if (j >= 9 && j < 12) exch[exch_shared_off(G, 3 + (j - 9)) + exch_pos<MODEL>(j)] = q4[3];
}
// Offset (doubles, always even = 16-B aligned) of the row a column reads as Mt: one of its window's exchange rows or
// one of the shared constant rows.
template <int MODEL>
CPI_HD int cov_row_off(int G, int g, int j) {
    const int r = cov_read_row<MODEL>(j);
    return (r < EXCH_GROUP_ROWS) ? g * EXCH_WIN + r * EXCH_PITCH : exch_shared_off(G, r - EXCH_GROUP_ROWS);
}

// j = column index inside the window's lane group: [0,NPCOL) covariance, [NPCOL,NCOL) transition, NCOL = idle.
template <int MODEL>
CPI_HD void cov_init(CovLane<MODEL> &L, int j, const double q4[4]) {
    typedef CovDims<MODEL> D;
#pragma unroll
    for (int i = 0; i < D::NR; i++) { L.P0[i] = 0; L.X[i] = 0; L.acc[i] = 0; }
    if (MODEL == 2 && j >= D::NPCOL) {
        // Discrete_J_b starts at identity (CpiV2.h:49): columns b_w (rows 3:6), b_a (9:12); the theta_klin columns
        // (18:21) have their unit entry outside the carried rows -- it enters through h (cov_h_offset)
        const int d = j - D::NPCOL;  // runtime per lane: selects only, no dynamically indexed registers
        const int hot = (d < 3) ? 3 + d : ((d < 6) ? 9 + (d - 3) : -1);
#pragma unroll
        for (int i = 0; i < D::NR; i++) L.P0[i] = (i == hot) ? 1.0 : 0.0;
    }
    // G Qc G^T = blkdiag(s_w^2, s_wb^2, s_a^2, s_ab^2, 0) (x) I  (CpiV1.h:283-291; Rs^T Rs = I)
    const double ht = 0.5 * q4[0], hv = 0.5 * q4[2];
    L.hqt = mk(j == 0 ? ht : 0.0, j == 1 ? ht : 0.0, j == 2 ? ht : 0.0);
    L.hqv = mk(j == 6 ? hv : 0.0, j == 7 ? hv : 0.0, j == 8 ? hv : 0.0);
}

// Start of an interval: pick up the shared per-interval vectors (ir = the interval record, hoff = cov_h_offset).
template <int MODEL>
CPI_HD void cov_begin(CovLane<MODEL> &L, const double *ir, int hoff) {
    typedef IrL<MODEL> IR;
    L.dt = ir[IR::DT];
    L.w = rec_v3(ir, IR::W);
    L.a = rec_v3(ir, IR::A);
    if (MODEL == 2) { L.gtau = rec_v3(ir, IR::GTAU); L.h = rec_v3(ir, hoff); }
    else { L.gtau = mk(0, 0, 0); L.h = mk(0, 0, 0); }
}

// Stage s: M = rows (theta, v, p) of F x for this lane's column (+ half its own diagonal process noise).
// Classic RK4 stage rotations R_old, R_mid, R_mid, R_new (CpiV1.h:279,300,332) come from the record (Rs).
// The stage rotation of the record: R_old, R_mid, R_mid, R_new.
template <int MODEL>
CPI_HD M3 cov_stage_rotation(const double *ir, int s) {   // stages 1..3 (stage 0: see IrL)
    typedef IrL<MODEL> IR;
    return rec_mat(ir, (s == 3) ? IR::RNEW : IR::RMID);
}
template <int MODEL>
CPI_HD void cov_stage_M(const CovLane<MODEL> &L, int s, const M3 &Rs, double M[9]) {
    const double *X = (s == 0) ? L.P0 : L.X;   // s is a compile-time constant after unrolling
    const V3 xt = mk(X[0], X[1], X[2]);
    const V3 xbw = mk(X[3], X[4], X[5]);
    const V3 xba = mk(X[9], X[10], X[11]);
    const V3 mt = (L.hqt - cross(L.w, xt)) - xbw;
    V3 y = cross(L.a, xt) + xba;
    if (MODEL == 2) {
        constexpr int o = (CovDims<MODEL>::NR >= 18) ? 15 : 0;
        const V3 xc = mk(X[o], X[o + 1], X[o + 2]);
        y = y + cross(L.gtau, xc) + L.h;
    }
    const V3 mv = L.hqv - mulT(Rs, y);
    M[0] = mt.x; M[1] = mt.y; M[2] = mt.z;
    M[3] = mv.x; M[4] = mv.y; M[5] = mv.z;
    M[6] = X[6]; M[7] = X[7]; M[8] = X[8];
}

// Finish stage s: k = M (rows theta,v,p) + Mt, then X <- P0 + c k and acc += wgt k.  Mt = the row of the
// exchange buffer selected by cov_read_row (the transposed F X row, a constant noise row, or zeros), so no
// per-lane condition is left in the arithmetic.
// Rows p of F X are rows v of X (F_pv = I), and X is symmetric: the transposed contribution a p column needs --
// row p_k of F X over all columns -- IS column v_k of X, i.e. the registers of the lane that owns v_k.  Where the lane map
// puts the p lanes in one 4-lane DPP bank and the v lanes a fixed distance below (model 1: lanes 12-14 <- 6-8), the
// kernel takes them with masked row_shr DPP moves instead of sending 3 of the 9 exchange rows through LDS (the LDS pipe,
// shared by the 8 wavefronts of a CU, is what bounds the covariance kernels; the VALU has slack).
template <int MODEL> struct CovPBySymmetry { static const bool V = (MODEL == 1) || (CPI_COV2_PSYM != 0); };
// the p lanes of a window group and the distance to their v lanes (same DPP row): model 1 lanes 12-15 <- 6-9, model 2 (lane
// map above) lanes 28-31 <- 22-25
template <int MODEL> struct CovPLanes { static const int FIRST = (MODEL == 1) ? 12 : 28, SHIFT = 6; };
template <int MODEL> struct CovExchRows { static const int V = CovPBySymmetry<MODEL>::V ? 6 : 9; };   // rows written per stage
template <int MODEL>
CPI_HD const double *cov_stage_X(const CovLane<MODEL> &L, int s) { return (s == 0) ? L.P0 : L.X; }

// mt[i] = the transposed contribution to row i (already picked out of the exchange row / taken by symmetry)
template <int MODEL>
CPI_HD void cov_stage_finish_regs(CovLane<MODEL> &L, int s, const double M[9], const double *mt) {
    typedef CovDims<MODEL> D;
    const double dt = L.dt;
    const double c = (s == 2) ? dt : 0.5 * dt;               // X for the next stage
    const double wgt = (s == 0 || s == 3) ? dt * (1.0 / 6.0) : dt * (1.0 / 3.0);
#pragma unroll
    for (int i = 0; i < D::NR; i++) {
        double k = mt[i];
        if (i < 3) k += M[i];
        else if (i >= 6 && i < 9) k += M[i - 3];
        else if (i >= 12 && i < 15) k += M[i - 6];
        if (s == 0) L.acc[i] = fma(wgt, k, L.P0[i]);
        else if (s < 3) L.acc[i] = fma(wgt, k, L.acc[i]);
        if (s < 3) L.X[i] = fma(c, k, L.P0[i]);
        else L.P0[i] = fma(wgt, k, L.acc[i]);
    }
}
template <int MODEL>
CPI_HD void cov_stage_finish(CovLane<MODEL> &L, int s, const double M[9], const double *Mt) {
    typedef CovDims<MODEL> D;
    const double dt = L.dt;
    const double c = (s == 2) ? dt : 0.5 * dt;               // X for the next stage
    const double wgt = (s == 0 || s == 3) ? dt * (1.0 / 6.0) : dt * (1.0 / 3.0);
#pragma unroll
    for (int i = 0; i < D::NR; i++) {
        double k = Mt[exch_pos<MODEL>(i)];
        if (i < 3) k += M[i];
        else if (i >= 6 && i < 9) k += M[i - 3];
        else if (i >= 12 && i < 15) k += M[i - 6];
        // stage 0 starts the running sum from P0; stage 3 writes the finished column straight back into P0
        if (s == 0) L.acc[i] = fma(wgt, k, L.P0[i]);
        else if (s < 3) L.acc[i] = fma(wgt, k, L.acc[i]);
        if (s < 3) L.X[i] = fma(c, k, L.P0[i]);
        else L.P0[i] = fma(wgt, k, L.acc[i]);
    }
}

// End of interval: model 2 row-clone theta -> theta_clone (B_k of CpiV2.h:436-443).  The column clone
// (columns 15:18 := columns 0:3) is a cross-lane copy done by the kernel.
template <int MODEL>
CPI_HD void cov_end(CovLane<MODEL> &L) {
    typedef CovDims<MODEL> D;
    if (MODEL == 2) {
        constexpr int o = (D::NR >= 18) ? 15 : 0;   // (model 1 never takes this branch)
        L.P0[o] = L.P0[0]; L.P0[o + 1] = L.P0[1]; L.P0[o + 2] = L.P0[2];
    }
}

// ------------------------------------------------------------------------------------------
// evaluateError (ImuFactorCPIv1.cpp:37-208 / ImuFactorCPIv2.cpp:38-212).
// What the factor constructors copy (ImuFactorCPIv1.h:78-100), by reference: every field stays in memory
// (LDS in the kernel) and is read where it is used, so nothing is held in registers across the evaluation.
// Matrices are column-major; lin = {bg_lin[3], ba_lin[3]}; xi/xj = JPLNavState [q(4) bg(3) v(3) ba(3) p(3)].
struct FactorMeas {
    const double *alpha, *beta, *q_KtoK1, *lin, *J_q, *J_beta, *J_alpha, *H_beta, *H_alpha, *dt, *q_K_lin, *O_beta, *O_alpha;
    const double *xi, *xj;
    V3 grav;
};
CPI_HD V3 ldv(const double *p) { return mk(p[0], p[1], p[2]); }
CPI_HD Q4 ldq(const double *p) { Q4 q; q.x = p[0]; q.y = p[1]; q.z = p[2]; q.w = p[3]; return q; }
CPI_HD V3 mulcm(const double *A, V3 v) {  // column-major 3x3 in memory times vector
    return mk(A[0] * v.x + A[3] * v.y + A[6] * v.z, A[1] * v.x + A[4] * v.y + A[7] * v.z, A[2] * v.x + A[5] * v.y + A[8] * v.z);
}
CPI_HD V3 colcm(const double *A, int cc) { return mk(A[cc * 3], A[cc * 3 + 1], A[cc * 3 + 2]); }
CPI_HD M3 ldcm(const double *A) {
    M3 r;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) r.m[i][j] = A[j * 3 + i];
    return r;
}
struct NavState { Q4 q; V3 bg, v, ba, p; };  // JPLNavState.h:62-66 (state prediction output)


CPI_HD double sel3(double a, double b, double c, int k) { return k == 0 ? a : (k == 1 ? b : c); }
// quat_2_Rot(q) x without forming the matrix: (2w^2-1) x - 2w (v x x) + 2 v (v.x)   (quat_ops.h:104-109)
CPI_HD V3 qrot(Q4 q, V3 x) {
    const V3 v = mk(q.x, q.y, q.z);
    const double c = 2 * q.w * q.w - 1;
    return axpy(2 * dot(v, x), v, axpy(-2 * q.w, cross(v, x), c * x));
}
// (q_w I + sgn [q_v]x) y
CPI_HD V3 qLmul(Q4 q, double sgn, V3 y) { return axpy(sgn, cross(mk(q.x, q.y, q.z), y), q.w * y); }
CPI_HD V3 pick5(V3 a0, V3 a1, V3 a2, V3 a3, V3 a4, int b) {
    return mk(b == 0 ? a0.x : (b == 1 ? a1.x : (b == 2 ? a2.x : (b == 3 ? a3.x : a4.x))),
              b == 0 ? a0.y : (b == 1 ? a1.y : (b == 2 ? a2.y : (b == 3 ? a3.y : a4.y))),
              b == 0 ? a0.z : (b == 1 ? a1.z : (b == 2 ? a2.z : (b == 3 ? a3.z : a4.z))));
}

// One lane's share of evaluateError: the residual component err[c] and column c (0..14) of the dense
// 15x15 H1 and H2.  A column of a block is "block times unit vector", so with u = e_(c mod 3) held as a
// runtime vector everything is 3-vector algebra -- no 3x3 temporaries, no select chains over matrices.
// Split in three steps (shared quaternions + residual, H1 column, H2 column) so that a kernel can retire
// each output before computing the next and keep its register footprint small.
// Block layout: ImuFactorCPIv1.cpp:109-143 (H1), :169-185 (H2); model 2 adds the O_beta / O_alpha terms of
// ImuFactorCPIv2.cpp:73,75,115-119.
struct FactorShared {
    Q4 q_n, q_m, q_rminus, q_r, q_kR;
    V3 Ra, Rb, rku, u;
    double err_c;
    int bc, cc;
};
// Column-independent part: the five quaternions, R_k (p_j - p_i - ...), R_k (v_j - v_i ...) and the five
// 3-vector residual blocks e[0..4] = [2 q_r,vec ; b_g,K+1 - b_g,K ; betahat - beta ; b_a,K+1 - b_a,K ; alphahat - alpha].
// Round 6: every step's results are PINNED before the next step's loads are formed (CPI_PIN3 above tells why: the scheduling
// fences order the record's LDS loads but the arithmetic floats below the last of them, so the whole record -- ~100 doubles -- sat in
// registers at once).  Same arithmetic, same bits; registers of the sweeps that inline this function: dense 188 -> 90-96 (model 2:
// 232 -> 120-128), packed 188 -> 96, whitened 194 / 238 -> 144 / 151, Hessian 190 / 238 -> 128.  What that buys is occupancy where
// LDS allows it -- the whitened sweep runs three wavefronts per SIMD now (cpi_factor_kernels.hpp: CPI_FACTOR_W3; 1.26 -> 1.17 ms
// per 1 M factors, with R packed 1.10) -- and nothing where LDS (packed sweep: 5 wavefronts per CU, Hessian: 9) or HBM (dense sweep)
// is the limit: profiles/r06_packed.md.  -DCPI_CORE_PIN=0 restores the unpinned core for A/B runs.
#ifndef CPI_CORE_PIN
#define CPI_CORE_PIN 1
#endif
#if CPI_CORE_PIN && defined(__HIP_DEVICE_COMPILE__)
#define CPI_CORE_PINV(v) CPI_PIN3((v).x, (v).y, (v).z)
#define CPI_CORE_PINQ(q) do { asm volatile("" : "+v"((q).x), "+v"((q).y), "+v"((q).z), "+v"((q).w) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CPI_CORE_PINV(v) ((void)0)
#define CPI_CORE_PINQ(q) ((void)0)
#endif
template <int MODEL>
CPI_HD void factor_shared_core(const FactorMeas &f, FactorShared &S, V3 e[5]) {
    V3 dbg = ldv(f.xi + 4) - ldv(f.lin), dba = ldv(f.xi + 10) - ldv(f.lin + 3);
    CPI_CORE_PINV(dbg); CPI_CORE_PINV(dba);
    V3 ja = mulcm(f.J_alpha, dbg) + mulcm(f.H_alpha, dba);      // alpha corrections
    CPI_CORE_PINV(ja);
    CPI_SCHED_FENCE();
    V3 jb = mulcm(f.J_beta, dbg) + mulcm(f.H_beta, dba);        // beta corrections
    CPI_CORE_PINV(jb);
    CPI_SCHED_FENCE();
    const Q4 qi = ldq(f.xi);
    S.q_kR.x = 0; S.q_kR.y = 0; S.q_kR.z = 0; S.q_kR.w = 1;
    if (MODEL == 2) {
        S.q_kR = quat_multiply(qi, quat_inv(ldq(f.q_K_lin)));
        const V3 dthk = mk(2 * S.q_kR.x, 2 * S.q_kR.y, 2 * S.q_kR.z);
        ja = ja + mulcm(f.O_alpha, dthk);
        jb = jb + mulcm(f.O_beta, dthk);
        CPI_CORE_PINV(ja); CPI_CORE_PINV(jb); CPI_CORE_PINQ(S.q_kR);
        CPI_SCHED_FENCE();
    }
    const double dt = f.dt[0];
    {
        const V3 vi = ldv(f.xi + 7);
        V3 pa = (ldv(f.xj + 13) - ldv(f.xi + 13)) - dt * vi;
        V3 pb = ldv(f.xj + 7) - vi;
        if (MODEL == 1) { pa = pa + (0.5 * dt * dt) * f.grav; pb = pb + dt * f.grav; }
        S.Ra = qrot(qi, pa); S.Rb = qrot(qi, pb);
        CPI_CORE_PINV(S.Ra); CPI_CORE_PINV(S.Rb);
    }
    e[4] = (S.Ra - ja) - ldv(f.alpha);   // alphahat - alpha
    e[2] = (S.Rb - jb) - ldv(f.beta);    // betahat - beta
    CPI_CORE_PINV(e[4]); CPI_CORE_PINV(e[2]);
    CPI_SCHED_FENCE();
    const Q4 q_meas = ldq(f.q_KtoK1);
    Q4 q_b = rot_2_quat(Exp_so3(-(mulcm(f.J_q, dbg))));
    CPI_CORE_PINQ(q_b);
    S.q_n = quat_multiply(ldq(f.xj), quat_inv(qi));
    S.q_rminus = quat_multiply(S.q_n, quat_inv(q_meas));
    S.q_r = quat_multiply(S.q_rminus, q_b);
    S.q_m = quat_multiply(quat_inv(q_b), q_meas);
    e[0] = mk(2 * S.q_r.x, 2 * S.q_r.y, 2 * S.q_r.z);
    e[1] = ldv(f.xj + 4) - ldv(f.xi + 4);
    e[3] = ldv(f.xj + 10) - ldv(f.xi + 10);
}
// Select the column: u = e_(c mod 3), R_k u.
CPI_HD void factor_set_column(const FactorMeas &f, int c, FactorShared &S) {
    S.bc = c / 3; S.cc = c - 3 * S.bc;
    S.u = unit(S.cc);
    S.rku = qrot(ldq(f.xi), S.u);  // column cc of quat_2_Rot(q_GtoK)
}
template <int MODEL>
CPI_HD void factor_shared(const FactorMeas &f, int c, FactorShared &S) {
    V3 e[5];
    factor_shared_core<MODEL>(f, S, e);
    factor_set_column(f, c, S);
    const V3 ec = pick5(e[0], e[1], e[2], e[3], e[4], S.bc);
    S.err_c = sel3(ec.x, ec.y, ec.z, S.cc);
    CPI_SCHED_FENCE();
}
template <int MODEL>
CPI_HD void factor_H1_column(const FactorShared &S, const FactorMeas &f, double h1[15]) {
    const V3 u = S.u, z = mk(0, 0, 0);
    const int bc = S.bc, cc = S.cc;
    const V3 qnv = mk(S.q_n.x, S.q_n.y, S.q_n.z), qmv = mk(S.q_m.x, S.q_m.y, S.q_m.z);
    const V3 tt = -(qLmul(S.q_n, -1.0, qLmul(S.q_m, -1.0, u)) + dot(qmv, u) * qnv);         // (0,0)
    const V3 tg = qLmul(S.q_rminus, -1.0, colcm(f.J_q, cc));                                 // (0,3)
    put3(h1 + 0, pick5(tt, tg, z, z, z, bc));
    put3(h1 + 3, (bc == 1) ? -u : z);
    put3(h1 + 9, (bc == 3) ? -u : z);
    CPI_SCHED_FENCE();
    V3 vt = cross(S.Rb, u), pt = cross(S.Ra, u);                                             // (6,0) (12,0)
    if (MODEL == 2) {
        const V3 Lu = qLmul(S.q_kR, +1.0, u);
        vt = vt - mulcm(f.O_beta, Lu);
        pt = pt - mulcm(f.O_alpha, Lu);
    }
    CPI_SCHED_FENCE();
    {
        const V3 jb = colcm(f.J_beta, cc), hb = colcm(f.H_beta, cc);
        put3(h1 + 6, pick5(vt, -jb, -S.rku, -hb, z, bc));
    }
    CPI_SCHED_FENCE();
    {
        const V3 ja = colcm(f.J_alpha, cc), ha = colcm(f.H_alpha, cc);
        put3(h1 + 12, pick5(pt, -ja, -(f.dt[0] * S.rku), -ha, -S.rku, bc));
    }
}
// H2, column c: blkdiag(q_r,w I + [q_r,v]x, I, Rk, I, Rk)
CPI_HD void factor_H2_column(const FactorShared &S, double h2[15]) {
    const V3 u = S.u, z = mk(0, 0, 0);
    const int bc = S.bc;
    put3(h2 + 0, (bc == 0) ? qLmul(S.q_r, +1.0, u) : z);
    put3(h2 + 3, (bc == 1) ? u : z);
    put3(h2 + 6, (bc == 2) ? S.rku : z);
    put3(h2 + 9, (bc == 3) ? u : z);
    put3(h2 + 12, (bc == 4) ? S.rku : z);
}
template <int MODEL>
CPI_HD void factor_eval_col(const FactorMeas &f, int c, double &err_c, double h1[15], double h2[15]) {
    FactorShared S;
    factor_shared<MODEL>(f, c, S);
    err_c = S.err_c;
    factor_H1_column<MODEL>(S, f, h1);
    factor_H2_column(S, h2);
}

// ---- Hessian blocks of a factor by 3x3 BLOCK algebra (cpi_factor_hessian_kernel) ----------------
// What a GTSAM HessianFactor built from the linearised factor holds: M = [A1 A2 b]^T [A1 A2 b], A = R H, b = -R e
// (include/cpi_amd.h: cpi_factor_hessian_batch).  With Lam = R^T R (the information matrix P^-1) this is
//     M = Hc^T Lam Hc,   Hc = [H1 H2 -e]  (15 x 31),
// and the Jacobians are SPARSE in 3x3 blocks (ImuFactorCPIv1.cpp:109-143,169-185; rows / columns theta, b_g, v, b_a, p):
//     H1 = [ B    C     0      0     0  ]        H2 = blkdiag(A, I, Rk, I, Rk)
//          [ 0   -I     0      0     0  ]        B = H1(0,0)  C = H1(0,3)  E = H1(6,0)  F = H1(12,0)  A = H2(0,0)
//          [ E   -Jb   -Rk    -Hb    0  ]        Rk = R(q_GtoK); Jb, Ja, Hb, Ha = the measurement's bias Jacobians
//          [ 0    0     0     -I     0  ]
//          [ F   -Ja  -dt Rk  -Ha   -Rk ]
// so a row of Z = Lam H1 costs ten 3-vector x block products (90 FMAs) instead of 225, a column of G11 = H1^T Z another ten,
// the columns of G12 = (Lam H1)^T H2 and G22 = H2^T Lam H2 are combinations of three rows of Z / Lam with one column of a
// diagonal block of H2: ~3.5 k FMAs per factor where the dense route (31 whitened columns, 496 length-15 dot products)
// spends 11 k.
// ONE LANE PER ROW / COLUMN: lane q < 15 of a factor owns row q of Lam and of Z, then packed columns q and 15 + q of the
// result; lane 15 owns column 30 (g, f), which is the SAME arithmetic applied to -y = -Lam e instead of a column of Z.  The
// exchange (rows -> columns) goes through the arrays lam / zx (LDS on the device, host memory in tests/hostsim): 15 rows
// pitched 18 doubles, column 15 of zx holds y.  The functions below are a lane's arithmetic on a block table `blk` in memory.
// The device kernel shares state_blocks_column, h2_diag_col, rows_comb and pk with the host emulation; its products with table
// entries are DPP broadcast operands (cpi_factor_kernels.hpp: tab_h1t / tab_h2t, the table spread over the factor's lanes),
// of which z_row / h1t_vec / h2t_vec / lane_columns here are the plain-memory twins that tests/hostsim checks against the
// dense definition (same terms; sums of block products accumulate in one chain there, pairwise here).
namespace hsn {
static const int B_B = 0, B_C = 9, B_E = 18, B_F = 27, B_A = 36, B_RK = 45, B_JB = 54, B_JA = 63, B_HB = 72, B_HA = 81,
                 B_ERR = 90, B_DT = 105, BLK_D = 108;       // 3x3 blocks ROW-major; residual e[15]; dt; 2 of padding
static const int ROWP = 18, MAT_D = 15 * ROWP;              // lam / zx: 15 rows pitched 18 doubles = 36 dwords.  Rows stay 16-byte aligned, the
                                                            // sixteen lanes of a factor WRITE their rows to sixteen different 4-bank groups (q * 36 mod
                                                            // 64: pitch 16 put them on two -- an 8-way conflict), the five block rows a column phase reads
                                                            // start on different banks, and the four factors of a wavefront (270 doubles = 28 mod 64 apart) too
static const int OUT_D = 496;

CPI_HD M3 ldb(const double *p) {
    M3 A;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) A.m[r][c] = p[r * 3 + c];
    return A;
}
CPI_HD void stb(double *p, const M3 &A) {
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) p[r * 3 + c] = A.m[r][c];
}
// The state-dependent blocks, one column at a time: column cc of B, C, E, F, A and Rk (what factor_H1_column /
// factor_H2_column put into columns cc and 3 + cc of H1 and column cc of H2), written into the row-major block table.
template <int MODEL>
CPI_HD void state_blocks_column(const FactorShared &S0, const FactorMeas &f, int cc, double *blk) {
    const V3 u = unit(cc);
    const V3 rku = qrot(ldq(f.xi), u);
    const V3 qnv = mk(S0.q_n.x, S0.q_n.y, S0.q_n.z), qmv = mk(S0.q_m.x, S0.q_m.y, S0.q_m.z);
    const V3 tt = -(qLmul(S0.q_n, -1.0, qLmul(S0.q_m, -1.0, u)) + dot(qmv, u) * qnv);       // H1(0,0)
    const V3 tg = qLmul(S0.q_rminus, -1.0, colcm(f.J_q, cc));                               // H1(0,3)
    V3 vt = cross(S0.Rb, u), pt = cross(S0.Ra, u);                                          // H1(6,0), H1(12,0)
    if (MODEL == 2) {
        const V3 Lu = qLmul(S0.q_kR, +1.0, u);
        vt = vt - mulcm(f.O_beta, Lu);
        pt = pt - mulcm(f.O_alpha, Lu);
    }
    const V3 ta = qLmul(S0.q_r, +1.0, u);                                                   // H2(0,0)
    const V3 col[6] = { tt, tg, vt, pt, ta, rku };
    const int at[6] = { B_B, B_C, B_E, B_F, B_A, B_RK };
#pragma unroll
    for (int k = 0; k < 6; k++) { blk[at[k] + cc] = col[k].x; blk[at[k] + 3 + cc] = col[k].y; blk[at[k] + 6 + cc] = col[k].z; }
}
// row q of Lam = R^T R; R column-major 15 x 15, upper triangular WITH its zeros stored (the contract of
// cpi_sqrt_information_batch): Lam[q][c] = sum_{k <= c} R[k][q] R[k][c], the terms k > q vanish with R[k][q]
CPI_HD void lambda_row(const double *R, int q, double l[15]) {
    double own[15];
#pragma unroll
    for (int k = 0; k < 15; k++) own[k] = R[q * 15 + k];
#pragma unroll
    for (int c = 0; c < 15; c++) {
        double a = 0.0;
#pragma unroll
        for (int k = 0; k <= c; k++) a = fma(own[k], R[c * 15 + k], a);     // R[c * 15 + k]: the same address for the lanes of a factor
        l[c] = a;
    }
}
// v^T M as a vector (= M^T v), M a row-major block in memory
CPI_HD V3 vTm(V3 v, const double *M) { return mulT(ldb(M), v); }
// row q of Z = Lam H1 from row q of Lam, and y_q = (Lam e)_q
CPI_HD void z_row(const double l[15], const double *blk, double z[15], double &y) {
    const V3 lt = ldv(l), lg = ldv(l + 3), lv = ldv(l + 6), la = ldv(l + 9), lp = ldv(l + 12);
    put3(z + 0, vTm(lt, blk + B_B) + vTm(lv, blk + B_E) + vTm(lp, blk + B_F));
    CPI_PIN3(z[0], z[1], z[2]);
    put3(z + 3, (vTm(lt, blk + B_C) - lg) - (vTm(lv, blk + B_JB) + vTm(lp, blk + B_JA)));
    CPI_PIN3(z[3], z[4], z[5]);
    {
        const M3 Rk = ldb(blk + B_RK);
        put3(z + 6, -mulT(Rk, axpy(blk[B_DT], lp, lv)));
        put3(z + 12, -mulT(Rk, lp));
    }
    CPI_PIN3(z[6], z[7], z[8]);
    CPI_PIN3(z[12], z[13], z[14]);
    put3(z + 9, -((vTm(lv, blk + B_HB) + vTm(lp, blk + B_HA)) + la));
    CPI_PIN3(z[9], z[10], z[11]);
    const double *e = blk + B_ERR;
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < 15; c++) acc = fma(l[c], e[c], acc);
    y = acc;
}
// g = H1^T v for a 15-vector v (a column of Z: that column of G11; -y: the g1 part of column 30)
CPI_HD void h1t_vec(const double v[15], const double *blk, double g[15]) {
    const V3 vt = ldv(v), vg = ldv(v + 3), vv = ldv(v + 6), va = ldv(v + 9), vp = ldv(v + 12);
    put3(g + 0, mulT(ldb(blk + B_B), vt) + mulT(ldb(blk + B_E), vv) + mulT(ldb(blk + B_F), vp));
    CPI_PIN3(g[0], g[1], g[2]);
    put3(g + 3, (mulT(ldb(blk + B_C), vt) - vg) - (mulT(ldb(blk + B_JB), vv) + mulT(ldb(blk + B_JA), vp)));
    CPI_PIN3(g[3], g[4], g[5]);
    {
        const M3 Rk = ldb(blk + B_RK);
        put3(g + 6, -mulT(Rk, axpy(blk[B_DT], vp, vv)));
        put3(g + 12, -mulT(Rk, vp));
    }
    CPI_PIN3(g[6], g[7], g[8]);
    CPI_PIN3(g[12], g[13], g[14]);
    put3(g + 9, -((mulT(ldb(blk + B_HB), vv) + mulT(ldb(blk + B_HA), vp)) + va));
    CPI_PIN3(g[9], g[10], g[11]);
}
// column n of the diagonal block D_j of H2 (A, I, Rk, I, Rk)
CPI_HD V3 h2_diag_col(const double *blk, int j, int n) {
    const double *Ab = blk + B_A, *Rb = blk + B_RK;
    const V3 a = mk(Ab[n], Ab[3 + n], Ab[6 + n]), r = mk(Rb[n], Rb[3 + n], Rb[6 + n]), u = unit(n);
    return (j == 0) ? a : (((j & 1) != 0) ? u : r);
}
// out[c] = sum_m d[m] rows[(3 j + m) * ROWP + c]: column n of (block row j of a matrix)^T D_j -- applied to Z: a column
// of G12; applied to Lam: the vector w below
CPI_HD void rows_comb3(const double *r0, const double *r1, const double *r2, V3 d, double out[15]) {
#pragma unroll
    for (int c0 = 0; c0 < 15; c0 += 5) {      // five columns at a time: 15 doubles of rows in flight, not 45
#pragma unroll
        for (int c = c0; c < c0 + 5; c++) out[c] = fma(d.z, r2[c], fma(d.y, r1[c], d.x * r0[c]));
        CPI_PIN3(out[c0], out[c0 + 1], out[c0 + 2]);      // (without the pins: the same registers, the same 1.39 ms)
        CPI_PIN1(out[c0 + 3]); CPI_PIN1(out[c0 + 4]);
    }
}
CPI_HD void rows_comb(const double *mat, int j, V3 d, double out[15]) {
    rows_comb3(mat + (3 * j) * ROWP, mat + (3 * j + 1) * ROWP, mat + (3 * j + 2) * ROWP, d, out);
}
// t[3 i + m] = (D_i^T w_i)[m]: from w = (D_j^T Lam_j.)[n, :] the column of G22; from w = -y the g2 part of column 30
CPI_HD void h2t_vec(const double w[15], const double *blk, double t[15]) {
    put3(t + 0, mulT(ldb(blk + B_A), ldv(w)));
    CPI_PIN3(t[0], t[1], t[2]);
    put3(t + 3, ldv(w + 3));
    const M3 Rk = ldb(blk + B_RK);
    put3(t + 6, mulT(Rk, ldv(w + 6)));
    put3(t + 9, ldv(w + 9));
    put3(t + 12, mulT(Rk, ldv(w + 12)));
    CPI_PIN3(t[6], t[7], t[8]);
    CPI_PIN3(t[12], t[13], t[14]);
}
// position of entry (r, d), r <= d, in the packed upper triangle
CPI_HD int pk(int r, int d) { return r + d * (d + 1) / 2; }
// Everything lane q (0 .. 15) stores, from the exchange arrays: g[15] = rows 0 .. 14 of packed column q (q = 15: of column
// 30), u[15] = rows 0 .. 14 of packed column 15 + q (unused for q = 15), t[15] = rows 15 .. 29 of that column (q = 15: of
// column 30), f = entry (30, 30) (meaningful for q = 15 only).  Of g and t a lane q < 15 owns the entries r <= q.
CPI_HD void lane_columns(int q, const double *lam, const double *zx, const double *blk, double g[15], double u[15], double t[15], double &f) {
    const int j = (q < 15) ? q / 3 : 0, n = (q < 15) ? q - 3 * j : 0;
    const double sgn = (q == 15) ? -1.0 : 1.0;
    // Three independent results, fenced apart, the one with the largest temporaries (three blocks of H1 at a time) FIRST,
    // while nothing else is live: computed last it pushed the kernel past 256 registers (476 bytes of scratch per lane, and
    // scratch at this scale is HBM traffic: 10 GB per million factors with the first version).
    {
        double zc[15];
#pragma unroll
        for (int k = 0; k < 15; k++) zc[k] = sgn * zx[k * ROWP + q];       // column q of Z; for q = 15: -y
        h1t_vec(zc, blk, g);
        const double *e = blk + B_ERR;
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < 15; c++) acc = fma(e[c], zc[c], acc);
        f = -acc;
        CPI_PIN1(f);
    }
    const V3 d = h2_diag_col(blk, j, n);
    {
        double w[15];
        rows_comb(lam, j, d, w);
#pragma unroll
        for (int c = 0; c < 15; c++) w[c] = (q == 15) ? -zx[c * ROWP + 15] : w[c];      // lane 15: w = -y
        h2t_vec(w, blk, t);
    }
    rows_comb(zx, j, d, u);
}
}  // namespace hsn

// State prediction (GraphSolver_IMU.cpp:263-281 / 289-307).
template <int MODEL>
CPI_HD NavState predict_state(const NavState &xi, V3 alpha, V3 beta, Q4 q_KtoK1, double dt, V3 grav) {
    NavState o;
    o.q = quat_multiply(q_KtoK1, xi.q);
    const M3 Rinv = quat_2_Rot(quat_inv(xi.q));
    const V3 rb = mul(Rinv, beta), ra = mul(Rinv, alpha);
    o.bg = xi.bg; o.ba = xi.ba;
    if (MODEL == 1) {
        o.v = (xi.v - dt * grav) + rb;
        o.p = ((xi.p + dt * xi.v) - (0.5 * dt * dt) * grav) + ra;
    } else {
        o.v = xi.v + rb;
        o.p = (xi.p + dt * xi.v) + ra;
    }
    return o;
}

// ==========================================================================================
// Forster / GTSAM discrete preintegration comparator: what
// GraphSolver::createimufactor_discrete (GraphSolver_IMU.cpp:141-232) obtains from GTSAM's
// PreintegratedCombinedMeasurements (ManifoldPreintegration; GTSAM is absent from the reference tree --
// the algorithm is restated from its publication, see oracle/forster_oracle.c: PARITY UNPINNED).
//
// Per IMU interval, with w = w_m - b_g, a = a_m - b_a, E = Exp(w dt), Jr = right Jacobian of Exp at w dt:
//   means (NavState::update)        R' = R E,  p' = p + R (dt R^T v + dt^2/2 a),  v' = v + R (dt a)
//   bias Jacobians (ManifoldPreintegration::update), column c of each 3x3:
//       D = -R [a]x delRdelBg ;  delRdelBg' = E^T delRdelBg - Jr dt
//       delPdelBa += delVdelBa dt - dt^2/2 R ;  delPdelBg += delVdelBg dt + dt^2/2 D
//       delVdelBa += -R dt ;                    delVdelBg += D dt
//   covariance (CombinedImuFactor), P' = F P F^T + G, errors of p and v expressed in the body frame
//   (NavState::retract), written here directly in the order [theta b_g v b_a p] the call site swaps it into:
//       (F x)_theta = E^T x_theta - Jr dt x_bg
//       (F x)_v     = E^T (x_v - dt ([a]x x_theta + x_ba))
//       (F x)_p     = E^T (x_p + dt x_v - dt^2/2 [a]x x_theta)      -- no b_a term: the GTSAM the reference pins (c21186c6, 4.0
//                     era) sets only theta_H_biasOmega and vel_H_biasAcc in CombinedImuFactor ("TODO: should we not also account
//                     for bias on position?"); pos_H_biasAcc was added to GTSAM years later (round-2 advisor finding)
//       (F x)_bg = x_bg, (F x)_ba = x_ba
//       G = diag( (1/dt) (Jr dt) s_w^2 (Jr dt)^T, dt s_wb^2 I, dt s_a^2 I, dt s_ab^2 I, 0 )
// F does not depend on the running means, so the per-interval records are independent of each other.
namespace fsd {
// interval record (doubles): dt, s_w^2/dt, a, -, E (row-major), Jr*dt (row-major)
static const int IR_DT = 0, IR_QS = 1, IR_A = 2, IR_E = 6, IR_JD = 15, IR_SIZE = 24;

struct Rec { double dt, qs; V3 a; M3 E, JD; };

// One interval: reading (w_m, a_m) held over [t0, t1] (GraphSolver_IMU.cpp:171-180).  dt <= 0 (and NaN) gives the
// exact no-op record (GTSAM itself would divide by dt == 0; skipped like CpiV1.h:72-74, see forster_oracle.c).
CPI_HD Rec make_rec(double t0, double t1, V3 wm, V3 am, V3 bg, V3 ba, double wCov) {
    Rec r;
    double dt = t1 - t0;
    const bool live = dt > 0;
    if (!live) dt = 0;
    r.dt = dt;
    r.qs = live ? wCov / dt : 0.0;
    r.a = live ? am - ba : mk(0, 0, 0);
    const V3 phi = live ? dt * (wm - bg) : mk(0, 0, 0);
    double th, ith;
    mag_and_inverse(dot(phi, phi), th, ith);
    double s2, c2;
    sincos_fast(0.5 * th, s2, c2);
    const double sn = 2.0 * s2 * c2, omc = 2.0 * s2 * s2;   // sin(theta), 1 - cos(theta)
    const double ith2 = ith * ith;
    const double k1 = sn * ith, k2 = omc * ith2, k3 = (1.0 - k1) * ith2;
    r.E = poly_wx(phi, 1.0, k1, k2);            // I + sin/t [phi]x + (1-cos)/t^2 [phi]x^2
    const M3 Jr = poly_wx(phi, 1.0, -k2, k3);   // I - (1-cos)/t^2 [phi]x + (t - sin)/t^3 [phi]x^2
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.JD.m[i][j] = dt * Jr.m[i][j];
    return r;
}
CPI_HD void put_rec(double *o, const Rec &r) {
    o[IR_DT] = r.dt; o[IR_QS] = r.qs; put3(o + IR_A, r.a); o[IR_A + 3] = 0.0;
    rec_put_mat(o, IR_E, r.E); rec_put_mat(o, IR_JD, r.JD);
}
CPI_HD Rec get_rec(const double *o) {
    Rec r;
    r.dt = o[IR_DT]; r.qs = o[IR_QS]; r.a = rec_v3(o, IR_A);
    r.E = rec_mat(o, IR_E); r.JD = rec_mat(o, IR_JD);
    return r;
}

// A v + c and A^T v + c as pure FMA chains
CPI_HD V3 mul_acc(const M3 &A, V3 v, V3 c) {
    return mk(fma(A.m[0][0], v.x, fma(A.m[0][1], v.y, fma(A.m[0][2], v.z, c.x))),
              fma(A.m[1][0], v.x, fma(A.m[1][1], v.y, fma(A.m[1][2], v.z, c.y))),
              fma(A.m[2][0], v.x, fma(A.m[2][1], v.y, fma(A.m[2][2], v.z, c.z))));
}
CPI_HD V3 mulT_acc(const M3 &A, V3 v, V3 c) {
    return mk(fma(A.m[0][0], v.x, fma(A.m[1][0], v.y, fma(A.m[2][0], v.z, c.x))),
              fma(A.m[0][1], v.x, fma(A.m[1][1], v.y, fma(A.m[2][1], v.z, c.y))),
              fma(A.m[0][2], v.x, fma(A.m[1][2], v.y, fma(A.m[2][2], v.z, c.z))));
}
// a x b + c
CPI_HD V3 cross_acc(V3 a, V3 b, V3 c) {
    return mk(fma(a.y, b.z, fma(-a.z, b.y, c.x)), fma(a.z, b.x, fma(-a.x, b.z, c.y)), fma(a.x, b.y, fma(-a.y, b.x, c.z)));
}

struct Mean { M3 R; V3 p, v; double dT; };
CPI_HD void mean_init(Mean &m) { m.R = eye(); m.p = mk(0, 0, 0); m.v = mk(0, 0, 0); m.dT = 0; }
// NavState::update forms p' = p + R (dt R^T v + dt^2/2 a); R R^T = I to rounding, so the body-velocity round trip is
// skipped: p' = p + dt v + dt^2/2 (R a)  (differs from the literal form by O(1e-16 |v| dt); parity is unpinned and
// the gate is 1e-9).
CPI_HD void mean_step(Mean &m, const Rec &r) {
    const double dt22 = 0.5 * r.dt * r.dt;
    const V3 Ra = mul(m.R, r.a);
    m.p = axpy(dt22, Ra, axpy(r.dt, m.v, m.p));
    m.v = axpy(r.dt, Ra, m.v);
    m.R = mm(m.R, r.E);
    m.dT += r.dt;
}

// One column of the bias Jacobians per lane.  ek = e_c on a lane that owns accelerometer-bias column c (delVdelBa,
// delPdelBa; its r stays 0), eg = e_c on a lane that owns gyro-bias column c (delRdelBg, delVdelBg, delPdelBg);
// both zero on every other lane -- no selects in the recursion.
struct JacCol { V3 r, v, p; };
CPI_HD void jac_init(JacCol &J) { J.r = mk(0, 0, 0); J.v = mk(0, 0, 0); J.p = mk(0, 0, 0); }
CPI_HD void jac_step(JacCol &J, const M3 &Rold, const Rec &r, V3 ek, V3 eg) {
    const double dt22 = 0.5 * r.dt * r.dt;
    const V3 Ru = mul(Rold, cross_acc(r.a, J.r, ek));   // minus the column of D_acc_biasOmega, or of -R
    J.p = axpy(-dt22, Ru, axpy(r.dt, J.v, J.p));
    J.v = axpy(-r.dt, Ru, J.v);
    J.r = mulT_acc(r.E, J.r, -mul(r.JD, eg));
}

// y = F x for one covariance column (order [theta b_g v b_a p]); only the theta / v / p rows change.
CPI_HD void F_apply(const Rec &r, const double *x, double *y) {
    const V3 th = mk(x[0], x[1], x[2]), bg = mk(x[3], x[4], x[5]), v = mk(x[6], x[7], x[8]);
    const V3 ba = mk(x[9], x[10], x[11]), p = mk(x[12], x[13], x[14]);
    const double dt22 = 0.5 * r.dt * r.dt;
    const V3 cp = cross(r.a, th);      // position row: NO accelerometer-bias term (see the header comment: F(p, b_a) = 0 at c21186c6)
    const V3 c = cp + ba;              // velocity row: vel_H_biasAcc
    put3(y + 0, mulT_acc(r.E, th, -mul(r.JD, bg)));
    put3(y + 3, bg);
    put3(y + 6, mulT(r.E, axpy(-r.dt, c, v)));
    put3(y + 9, ba);
    put3(y + 12, mulT(r.E, axpy(-dt22, cp, axpy(r.dt, v, p))));
}
// x_theta += column j < 3 of G's theta block = (s_w^2/dt) (Jr dt) (Jr dt)^T e_j ; jdrow = row j of Jr dt, on = 1 on
// the lanes j < 3 and 0 elsewhere
CPI_HD void theta_noise_add(double *x, const Rec &r, V3 jdrow, double on) {
    const V3 g = mul_acc(r.JD, (r.qs * on) * jdrow, mk(x[0], x[1], x[2]));
    x[0] = g.x; x[1] = g.y; x[2] = g.z;
}
}  // namespace fsd

}  // namespace cpi
