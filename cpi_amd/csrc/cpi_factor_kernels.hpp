// cpi_factor_kernels.hpp -- evaluateError sweeps (dense, packed, whitened, Hessian blocks), square-root information, state prediction.
// Part of the translation unit cpi_factor.hip (included there after cpi_math.hpp / cpi_device_util.hpp; not a stand-alone header).
#pragma once

namespace {

// ============================================================================================
// factor kernels
// ============================================================================================
// (FactorArgs, PredictArgs: cpi_args.hpp)

__device__ __forceinline__ NavState ld_state(const double *p) {
    NavState s;
    s.q = ldq4(p); s.bg = ldv3(p + 4); s.v = ldv3(p + 7); s.ba = ldv3(p + 10); s.p = ldv3(p + 13);
    return s;
}

// One SoA input field (K doubles per factor) of the FPW consecutive factors of a wavefront: FPW*K contiguous
// doubles, lane i takes doubles i, i + 64, ...  load() is unconditional (index clamped to the last valid double),
// store() writes record-major into the LDS staging area.
// 16 bytes at an 8-byte aligned address with the non-temporal hint: the sweeps' outputs are written once and read by a later
// kernel (the solver), never by this one.  Round 4, same-box A/B per 1 M factors: square-root information 709-716 -> 674-680 us,
// Hessian blocks 1414 -> 1388 us (v2 1490 -> 1470), dense H1 / H2 793-804 -> 771-775 us on one box and unchanged on another,
// whitened 1271 -> 1250 us, packed sweep 292 -> 280 us (v2 327 -> 318); the same hint on the LOADS -- the records, R of the whitened /
// Hessian sweeps, P of the square-root information -- is neutral (within +-1 %): not used.
typedef double cpi_d2v __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void st16_nt(double *dst2, double a, double b) {
    cpi_d2v v; v.x = a; v.y = b;
    __builtin_nontemporal_store(v, reinterpret_cast<cpi_d2v *>(dst2));
}
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

// Workgroup -> group of factors.  CPI_FACTOR_XCD = 1: workgroup b runs on XCD b mod 8 (observed placement, for speed only), so
// "group (b mod 8) * ceil(groups / 8) + b / 8" gives every XCD one contiguous eighth of the inputs and outputs (the launchers pad
// the grid to a multiple of 8; surplus workgroups leave at once).  0: group = b.
#ifndef CPI_FACTOR_XCD
#define CPI_FACTOR_XCD 0
#endif
__device__ __forceinline__ long long factor_group_of_block(long long groups) {
#if CPI_FACTOR_XCD
    const long long per = (groups + 7) >> 3;
    const long long g = (long long)(blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    return ((long long)(blockIdx.x >> 3) < per && g < groups) ? g : -1;
#else
    (void)groups;
    return blockIdx.x;
#endif
}
inline unsigned factor_grid(long long groups) { return CPI_FACTOR_XCD ? (unsigned)(((groups + 7) >> 3) << 3) : (unsigned)groups; }

// Record layout of one factor in the LDS staging area (doubles).  Round 6: the fields only the column-independent core reads
// (factor_shared_core) come FIRST -- 73 doubles that are dead once the core is done -- and the fields the column phase still
// needs (J_q, O_beta, O_alpha, state_i) last: the packed sweep writes its 72-double result over the dead part of the factor's
// own record instead of into a stage of its own (LDS per factor 1 504 -> 928 bytes: 8 instead of 5 wavefronts per CU).
namespace fin {
constexpr int O_ALPHA = 0, O_BETA = 3, O_Q = 6, O_LIN = 10, O_JB = 16, O_JA = 25, O_HB = 34, O_HA = 43, O_DT = 52, O_QK = 53,
              O_XJ = 57, DEAD_AFTER_CORE = 73, O_JQ = 73, O_OB = 82, O_OA = 91, O_XI = 100, IN_D = 116;
}
// RD: doubles of R per factor when WHITEN -- 225 (dense column-major) or CPI_TRI_DOUBLES (the packed upper triangle)
template <int MODEL, int FPW, bool WHITEN, int RD = 225>
__device__ __forceinline__ void factor_fetch_inputs(const FactorArgs &A, long long f0, int nf, int lane, double *sIn,
                                                    double *sR) {
    using namespace fin;
    constexpr bool whiten = WHITEN;
    constexpr int HB = FPW * RD;
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
            for (int r = 0; r < (HB + 63) / 64; r++) R_[r] = A.sqrt_info[f0 * RD + min(lane + 64 * r, nf * RD - 1)];
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
// row I of R h for the whitening of cpi_factor_kernel (16 lanes per factor): the terms k = I .. 14 in that order, R[I][k] =
// register Rc[I] of lane k.  (dpp_fmac: defined with the square-root-information kernel below.)
template <int K> __device__ __forceinline__ void dpp_fmac(double &acc, double b, double x);
template <int I, int K>
__device__ __forceinline__ void whiten_row_dpp(const double (&Rc)[15], const double (&h)[15], double &acc) {
    if constexpr (K < 15) { dpp_fmac<K>(acc, Rc[I], h[K]); whiten_row_dpp<I, K + 1>(Rc, h, acc); }
}
template <int I>
__device__ __forceinline__ void whiten_col_dpp(const double (&Rc)[15], double (&h)[15]) {
    if constexpr (I < 15) {
        double acc = 0.0;
        whiten_row_dpp<I, I>(Rc, h, acc);
        h[I] = acc;                           // rows are finished top-down, so h[k], k > I, is still unwhitened
        whiten_col_dpp<I + 1>(Rc, h);
    }
}
#ifndef CPI_FACTOR_WPS
#define CPI_FACTOR_WPS 1
#endif
// TRI (WHITEN only): R arrives as its packed upper triangle (cpi_factor_eval_whitened_tri_batch): 120 instead of 225 doubles per
// factor to fetch and to park in LDS; entry (i, k), i <= k, at i + k (k + 1) / 2.  The arithmetic never touched the zeros the
// dense form stores below the diagonal, so the outputs are the same bits.
// The whitened 16-lane instantiations run THREE wavefronts per SIMD (round 6): with the pinned core (cpi_math.hpp: CPI_CORE_PIN) they
// need 144 / 151 registers, and with R parked inside the H stage (below) 11.4 KB of LDS -- 14 wavefronts per CU by LDS, 12 by
// registers.  Measured per 1 M factors, same box, alternating: dense R 1.26 -> 1.17 ms (model 2: 1.35 -> 1.26), packed R 1.22 -> 1.10
// (1.30 -> 1.19).  Without the pins the same bound spills 116 / 292 bytes per lane and LOSES (1.49 / 2.69 ms).  0 = two per SIMD, as before.
#ifndef CPI_FACTOR_W3
#define CPI_FACTOR_W3 1
#endif
#ifndef CPI_FACTOR_CORE_ONE_LANE
#define CPI_FACTOR_CORE_ONE_LANE 1
#endif
template <int MODEL, bool WHITEN, int LPF, bool TRI = false>
__global__ __launch_bounds__(64, (CPI_FACTOR_W3 && WHITEN && LPF == 16) ? 3 : CPI_FACTOR_WPS) void cpi_factor_kernel(FactorArgs A) {
    static_assert(WHITEN || !TRI, "TRI is a layout of the whitening matrix");
    constexpr int FPW = 64 / LPF;                // factors per wavefront
    constexpr int CPL = (15 + LPF - 1) / LPF;    // columns per lane
    constexpr int HB = FPW * 225;                // doubles of H1 (or H2) per wavefront
    constexpr int RD = TRI ? CPI_TRI_DOUBLES : 225;   // doubles of R per factor
    constexpr int IN_D = fin::IN_D;
    __shared__ __attribute__((aligned(16))) double sH[HB + FPW * 15 + 4];   // one 15x15 set at a time: H1, then H2
    __shared__ __attribute__((aligned(16))) double sIn[FPW * IN_D];         // the factors' input records
    // whitening only: the factors' R.  It is dead once the lanes hold their column of it (Rc) and R err is formed -- both happen
    // before the first column is written to the stage -- so it lives IN the stage (CPI_FACTOR_W3: 11.4 KB of LDS per wavefront)
    constexpr bool R_IN_STAGE = CPI_FACTOR_W3 && WHITEN && LPF == 16;
    __shared__ __attribute__((aligned(16))) double sR_own[(WHITEN && !R_IN_STAGE) ? FPW * RD : 2];
    double *sR = R_IN_STAGE ? sH : sR_own;
    const int lane = threadIdx.x;
    const int q = lane % LPF, fl = lane / LPF;
    const long long grp = factor_group_of_block((A.F + FPW - 1) / FPW);
    if (grp < 0) return;
    const long long f0 = grp * FPW;
    const int nf = (int)min((long long)FPW, A.F - f0);
    constexpr bool whiten = WHITEN;

    factor_fetch_inputs<MODEL, FPW, WHITEN, RD>(A, f0, nf, lane, sIn, sR);
    __syncthreads();
    const double *in = sIn + fl * IN_D;
    const FactorMeas m = factor_meas_of(in, A.grav);   // every field is read from LDS where it is used
    double *s1 = sH, *se = sH + HB;
    const double *Rf = sR + fl * RD;
    auto r_at = [&](int i, int k) -> double { return TRI ? Rf[k * (k + 1) / 2 + i] : Rf[k * 15 + i]; };   // R[i][k], i <= k
    const int nd = nf * 225, ne = nf * 15;

    // ---- shared algebra; the residual goes to the staging area at once.  Lane q of a factor publishes rows
    // q, q + LPF, ... of the 15-vector.
    FactorShared S;
    if constexpr (CPI_FACTOR_CORE_ONE_LANE != 0 && WHITEN && LPF == 16) {
        // ONE lane of a factor runs the core; the others pick its results up from LDS.  Same instructions for the wavefront (plus 13
        // 16-byte reads), 1 / 16 of the FP64 lanes switching during a third of the kernel: these sweeps run at the clock the power
        // controller leaves them (profiles/r06_small_sweeps.md section 6: whitened sweep with R packed 0.985 -> 0.925 ms per 1 M factors).
        // Not for the plain sweep (eight lanes per factor, bound by its 3.7 KB of output per factor): 2-3 % slower with it.  The results overwrite the parts of the factor's own record
        // that only the core reads in this kernel -- alpha, beta, q, lin (doubles 0-15) and q_K_lin, state_j (53-72); the bias Jacobians
        // in between are read again by factor_H1_column -- in-order DS: the core's reads are complete before its writes land.
        static_assert(fin::O_JB == 16 && fin::O_QK == 53 && fin::O_JQ >= 63, "the core's results overlay what only the core reads");
        double *rec = sIn + fl * IN_D;
        if (q == 0) {
            V3 e5[5];
            factor_shared_core<MODEL>(m, S, e5);
#pragma unroll
            for (int a = 0; a < 5; a++) put3(se + fl * 15 + 3 * a, e5[a]);
            const Q4 qs[5] = { S.q_n, S.q_m, S.q_rminus, S.q_r, S.q_kR };
#pragma unroll
            for (int a = 0; a < 5; a++) {
                double *o = rec + (a < 4 ? 4 * a : fin::O_QK);
                o[0] = qs[a].x; o[1] = qs[a].y; o[2] = qs[a].z; o[3] = qs[a].w;
            }
            put3(rec + fin::O_QK + 4, S.Ra); put3(rec + fin::O_QK + 7, S.Rb);
        }
        wave_lds_fence();
        S.q_n = ldq(rec); S.q_m = ldq(rec + 4); S.q_rminus = ldq(rec + 8); S.q_r = ldq(rec + 12); S.q_kR = ldq(rec + fin::O_QK);
        S.Ra = ldv(rec + fin::O_QK + 4); S.Rb = ldv(rec + fin::O_QK + 7);
    } else {
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
    // Sixteen lanes per factor = one DPP row: lane k keeps column k of R (Rc[i] = R[i][k]) and every R[i][k] h[k] is ONE
    // v_fmac_f64_dpp with lane k's register as the broadcast operand -- 240 LDS broadcasts per lane less (H1 and H2).
    double Rc[(WHITEN && LPF == 16) ? 15 : 1];
    if constexpr (WHITEN && LPF == 16) {
#pragma unroll
        for (int i = 0; i < 15; i++) Rc[i] = r_at(TRI ? min(i, min(q, 14)) : i, min(q, 14));   // (rows below the diagonal are never used: whiten_row_dpp)
    }
    auto whiten_col = [&](double (&h)[15]) {   // in place: out[i] = sum_{k >= i} R[i][k] h[k]
        if constexpr (WHITEN && LPF == 16) whiten_col_dpp<0>(Rc, h);
        else {
#pragma unroll
            for (int i = 0; i < 15; i++) {
                double acc = 0.0;
#pragma unroll
                for (int k = i; k < 15; k++) acc = fma(r_at(i, k), h[k], acc);
                h[i] = acc;                   // rows are finished top-down, so h[k], k > i, is still unwhitened
            }
        }
    };
    if (whiten) {   // R err needs the whole residual
        wave_lds_fence();
        double acc[CPL];
#pragma unroll
        for (int k = 0; k < CPL; k++) {
            const int cr = min(q + LPF * k, 14);
            acc[k] = 0.0;
            for (int j = 0; j < 15; j++) acc[k] = fma((j >= cr) ? r_at(cr, j) : 0.0, se[fl * 15 + j], acc[k]);
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
        for (int i = lane; i < n2; i += 64) st16_nt(dst + 2 * i, src[2 * i], src[2 * i + 1]);
        if ((n & 1) && lane == 0) dst[n - 1] = src[n - 1];
    };
    const Q4 qi = ldq(m.xi);
    if constexpr (R_IN_STAGE) wave_lds_fence();   // every read of R (Rc, R err) is issued before the first column lands on it
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
static_assert(FACTOR_PACKED_DOUBLES <= fin::DEAD_AFTER_CORE, "the packed result overlays the part of the record that only the core reads");
// What the column phase of the packed sweep needs of H1's column c: rows 0-2 (blocks (0,0) / (0,3)) and, for c < 3, rows 6-8 and 12-14
// (blocks (6,0), (12,0)) -- the terms of factor_H1_column without the rows the packed form derives from the measurement, so that
// nothing of the record's dead part (J_beta, H_beta, J_alpha, H_alpha, dt) is read after the result has begun to overwrite it.
template <int MODEL>
__device__ __forceinline__ void packed_H1_column(const FactorShared &S, const FactorMeas &f, V3 &top, V3 &vt, V3 &pt) {
    const V3 u = S.u;
    const V3 qnv = mk(S.q_n.x, S.q_n.y, S.q_n.z), qmv = mk(S.q_m.x, S.q_m.y, S.q_m.z);
    const V3 tt = -(qLmul(S.q_n, -1.0, qLmul(S.q_m, -1.0, u)) + dot(qmv, u) * qnv);         // (0,0)
    const V3 tg = qLmul(S.q_rminus, -1.0, colcm(f.J_q, S.cc));                               // (0,3)
    top = (S.bc == 0) ? tt : tg;
    vt = cross(S.Rb, u); pt = cross(S.Ra, u);                                                // (6,0) (12,0)
    if (MODEL == 2) {
        const V3 Lu = qLmul(S.q_kR, +1.0, u);
        vt = vt - mulcm(f.O_beta, Lu);
        pt = pt - mulcm(f.O_alpha, Lu);
    }
}
template <int MODEL, int LPF>
__global__ __launch_bounds__(64) void cpi_factor_packed_kernel(FactorArgs A, double *packed) {
    constexpr int FPW = 64 / LPF, PD = FACTOR_PACKED_DOUBLES, IN_D = fin::IN_D;
    __shared__ __attribute__((aligned(16))) double sIn[FPW * IN_D];
    __shared__ double sDummy[2];
    const int lane = threadIdx.x;
    const int q = lane % LPF, fl = min(lane / LPF, FPW - 1);   // 64 mod LPF spare lanes repeat the last factor's lane 0 (same values, same slots)
    const long long grp = factor_group_of_block((A.F + FPW - 1) / FPW);
    if (grp < 0) return;
    const long long f0 = grp * FPW;
    const int nf = (int)min((long long)FPW, A.F - f0);
    factor_fetch_inputs<MODEL, FPW, false>(A, f0, nf, lane, sIn, sDummy);
    __syncthreads();
    const double *in = sIn + fl * IN_D;
    const FactorMeas m = factor_meas_of(in, A.grav);
    double *out = sIn + fl * IN_D;                             // the result overlays the record's dead part (fin::DEAD_AFTER_CORE)
    FactorShared S;
    {
        V3 e5[5];
        factor_shared_core<MODEL>(m, S, e5);
        wave_lds_fence();                                      // the core's reads of every factor of the wavefront are issued: from here on the
                                                               // first 73 doubles of a record are result space
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
            S.bc = c / 3; S.cc = c - 3 * S.bc;
            S.u = unit(S.cc);
            S.rku = qrot(qi, S.u);
            V3 top, vt, pt;
            packed_H1_column<MODEL>(S, m, top, vt, pt);
            if (c < 3) {
                double *b = out + 15 + 3 * c;                    // H1(0,0), H1(6,0), H1(12,0): column c of each
                b[0] = top.x; b[1] = top.y; b[2] = top.z;
                b[9] = vt.x; b[10] = vt.y; b[11] = vt.z;
                b[18] = pt.x; b[19] = pt.y; b[20] = pt.z;
                const V3 h2 = qLmul(S.q_r, +1.0, S.u);           // H2(0,0) column c (factor_H2_column)
                double *r = out + 51 + 3 * c;                    // R(q_GtoK) column c, then H2(0,0) column c
                r[0] = S.rku.x; r[1] = S.rku.y; r[2] = S.rku.z;
                r[9] = h2.x; r[10] = h2.y; r[11] = h2.z;
            } else {
                double *b = out + 42 + 3 * (c - 3);              // H1(0,3) column c - 3
                b[0] = top.x; b[1] = top.y; b[2] = top.z;
            }
        }
    }
    if (q == LPF - 1) { out[69] = 0.0; out[70] = 0.0; out[71] = 0.0; }
    wave_lds_fence();
    // 36 pieces of 16 bytes per factor, consecutive lanes = consecutive pieces of the OUTPUT; piece i sits in record i / 36
    for (int i = lane; i < nf * (PD / 2); i += 64) {
        const int g = i / (PD / 2), r = i - g * (PD / 2);
        const double *src = sIn + g * IN_D + 2 * r;
        st16_nt(packed + f0 * PD + 2 * i, src[0], src[1]);
    }
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
    // gfx90a+ moves 64 bits at once when the control is a row broadcast: ONE v_mov_b64_dpp row_newbcast:K
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xf, 0xf, false);
}
// The VOP2 double-precision multiply-add of gfx90a+ takes a row broadcast as an operand modifier: acc += bcast_K(b) * x in
// ONE instruction.  hipcc does not fold a v_mov_b64_dpp into its user, hence the assembler text.  The hazard recogniser does
// not see through inline assembly ("a VALU write of a VGPR followed by a DPP read of it needs two wait states"): use these
// only with a broadcast source that comes out of memory / LDS or was pinned long before -- tests/test_abi.py
// (test_dpp_sources_are_not_fresh_valu_results) checks that rule on the disassembly of the shipped library.
template <int K>
__device__ __forceinline__ void dpp_fmac(double &acc, double b, double x) {      // acc += bcast_K(b) * x
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(x), "n"(K));
}
template <int K>
__device__ __forceinline__ void dpp_fnmac(double &acc, double b, double x) {     // acc -= bcast_K(b) * x: the DPP encoding carries source modifiers
    asm("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(x), "n"(K));
}
template <int K>
__device__ __forceinline__ double dpp_mul(double b, double x) {   // v_mul_f64 is VOP3-only (no DPP form): a chain starts from zero
    double r = 0.0;
    dpp_fmac<K>(r, b, x);
    return r;
}
// 1 / sqrt(a) for the pivots: the Newton sequence of mag_and_inverse (cpi_math.hpp) WITHOUT its clamp, so that a non-positive or
// NaN pivot poisons the factor by itself -- v_rsq_f64 gives NaN for a < 0 and +inf for 0, and 0 x inf = NaN carries it through the
// iteration -- where a clamp (v_max) plus a compare and two selects per step stood.  Same bits as before for a > 1e-280.
__device__ __forceinline__ double pivot_rsqrt(double a) { return inv_sqrt_unclamped(a); }
// No per-lane selects in the sweep (round 6: 763 -> see resource table; the packed form of this kernel is VALU-bound).  With
// acc[] started at -[i == j] instead of 0, row k of column j of U is -acc[k] / b_kk for EVERY lane and row:
//   k <  j : acc[k] = sum_m B[k][m] U[m][j], as before;
//   k == j : acc[j] is still -1 (every contribution to it carried the factor U[m][j] = 0, m > j): -(-1) / b_jj = 1 / b_jj;
//   k >  j : acc[k] is still 0: -0 / b_kk = -0 -- a zero (the dense store writes +0 below the diagonal explicitly), and
//            acc[i] += B[i][k] * (-0) leaves every running sum as it is.
// The trailing update of the working matrix needs no "finished lane" select either: a lane j >= k is never read again (the
// broadcasts of the remaining steps come from lanes < k), so whatever its dead registers turn into is nobody's input.
// -1.0 in the lanes of `mask` (a wave-uniform constant: it lives in a scalar register pair), 0.0 elsewhere
__device__ __forceinline__ double neg_delta_of_lane(unsigned long long mask) {
    int hi;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(hi) : "v"(0xBFF00000), "s"(mask));
    return __hiloint2double(hi, 0);
}
#ifndef CPI_SQRT_PIVOT_ONE_LANE
#define CPI_SQRT_PIVOT_ONE_LANE 1
#endif
#ifndef CPI_SQRT_SPLIT_MASK
#define CPI_SQRT_SPLIT_MASK 1     // packed form only (the dense form is bound by its 7.2 KB per factor: 1 % slower with it)
#endif
template <int K, bool SPLIT>
__device__ __forceinline__ void chol_inv_step(double (&a)[15], double (&u)[15], double (&acc)[15], int j) {
    if constexpr (K >= 0) {
        // pivot: 1 / b_kk, b_kk = sqrt(A[k][k]); a non-positive (or NaN) pivot poisons the factor with NaNs
#if CPI_SQRT_PIVOT_ONE_LANE
        // Lane k alone runs the eleven-instruction Newton sequence on ITS a[k] and the result is broadcast, instead of the argument
        // being broadcast and all sixteen lanes of the factor running it: the same instructions for the wavefront, a sixteenth of the
        // FP64 lanes switching during 29 % of the kernel -- these kernels run at the clock the power controller leaves them
        // (profiles/r06_small_sweeps.md section 6).
        double mine = a[K];
        if (j == K) mine = pivot_rsqrt(mine);
        asm volatile("s_nop 1" : "+v"(mine));    // a VALU result read as a DPP source: two wait states
        const double inv = row_share<K>(mine);
#else
        const double inv = pivot_rsqrt(row_share<K>(a[K]));
#endif
        u[K] = -acc[K] * inv;                    // row k of column j of U
        // trailing update of column j (all rows < k) with B[j][k] = A[k][j] / b_kk (symmetry: a static register of lane j).
        // B[i][k] = A[i][k] / b_kk sits in lane k: both updates take it as the broadcast operand of a DPP multiply-add, the
        // lane's own factors folded with 1 / b_kk first (two instructions per row where a broadcast, a multiply and two
        // multiply-adds stood).  Rows descend so that the pivot of the next step (a[K-1]) is the OLDEST write of this loop,
        // and the s_nop covers K = 1 -- inline assembly is invisible to the compiler's DPP hazard check in both directions.
        const double ca = -inv * (a[K] * inv), cu = inv * u[K];
        if constexpr (SPLIT) {
            // At step k a lane needs EITHER update: its running sums change only if its u[k] is non-zero (j >= k), its column of the
            // working matrix is read again only if it is still to be pivoted (j < k).  Two exec-masked loops -- lane k stays active in
            // both: it is the DPP source -- halve the FP64 lane activity of these 2 k instructions (the sums first: they read lane k's
            // column before lane k's own, dead, update overwrites it).  Steady state of 300-launch runs: packed 0.355 -> 0.336 ms per
            // 1 M factors = 0.71 of 8 TB/s, past the plain copy of its mix.
            if (j >= K) {
#pragma unroll
                for (int i = K - 1; i >= 0; i--) dpp_fmac<K>(acc[i], a[i], cu);
            }
            if (j <= K) {
#pragma unroll
                for (int i = K - 1; i >= 0; i--) dpp_fmac<K>(a[i], a[i], ca);
            }
        } else {
#pragma unroll
            for (int i = K - 1; i >= 0; i--) {
                dpp_fmac<K>(acc[i], a[i], cu);
                dpp_fmac<K>(a[i], a[i], ca);
            }
        }
        asm volatile("s_nop 1");
        __builtin_amdgcn_sched_barrier(0);   // keep the steps in order: hoisted broadcasts would cost ~200 registers
        chol_inv_step<K - 1, SPLIT>(a, u, acc, j);
    }
}
// PACKED (cpi_sqrt_information_packed_batch): P arrives as its upper triangle and R leaves as its non-zero triangle, 120 doubles
// each (CPI_TRI_INDEX).  Lane j rebuilds its full symmetric column from the triangle -- rows i <= j from column j, rows i > j
// from row j of column i -- and stores rows 0 .. j of its column of U: the same registers enter the same arithmetic as in the
// dense form.
// WPB wavefronts per workgroup, each on four factors of its own (no barrier between them: the kernel never leaves its wavefront).
#ifndef CPI_SQRT_WPB
#define CPI_SQRT_WPB 1
#endif
template <bool PACKED>
__global__ __launch_bounds__(64 * CPI_SQRT_WPB, 3) void cpi_sqrt_info_kernel(long long F, const double *P, double *Rout) {
    constexpr int FPW = 4, MD = PACKED ? CPI_TRI_DOUBLES : 225, WPB = CPI_SQRT_WPB;
    __shared__ __attribute__((aligned(16))) double sAll[WPB * FPW * MD];
    double *sA = sAll + (WPB > 1 ? (threadIdx.x >> 6) * (FPW * MD) : 0);
    const int lane = threadIdx.x & 63, j = lane & 15, fl = lane >> 4;
    const long long groups = (F + FPW - 1) / FPW;
    const long long grp = (WPB > 1) ? (long long)blockIdx.x * WPB + (threadIdx.x >> 6) : factor_group_of_block(groups);
    if (grp < 0 || grp >= groups) return;
    const long long f0 = grp * FPW;
    const int nf = (int)min((long long)FPW, F - f0);
    struct __attribute__((packed, aligned(8))) d2u { double a, b; };
    {
        const int n2 = (nf * MD) >> 1;
        const d2u *src = reinterpret_cast<const d2u *>(P + f0 * MD);
        for (int i = lane; i < n2; i += 64) { const d2u v = src[i]; sA[2 * i] = v.a; sA[2 * i + 1] = v.b; }
        if (((nf * MD) & 1) && lane == 0) sA[nf * MD - 1] = P[f0 * MD + nf * MD - 1];
    }
    wave_lds_fence();
    const int fc = min(fl, nf - 1), jc = min(j, 14);   // idle lanes (j == 15, missing factors) redo a valid column
    double a[15], u[15], acc[15];
    // (the packed form is bound by its instruction count: one select per element between two addresses whose constant parts are
    //  instruction offsets -- rows i <= j of the lane's column sit at T(j) + i, the others at T(i) + j -- and the -[i == j] start of
    //  the running sums from a CONSTANT lane mask, lanes i, 16 + i, 32 + i, 48 + i, instead of a compare per element)
    const double *colp = sA + fc * MD + jc * (jc + 1) / 2, *rowp = sA + fc * MD + jc;
#pragma unroll
    for (int i = 0; i < 15; i++) {
        if constexpr (PACKED) a[i] = ((i <= jc) ? colp + (i - i * (i + 1) / 2) : rowp)[i * (i + 1) / 2];
        else a[i] = sA[fc * MD + jc * 15 + i];
        acc[i] = neg_delta_of_lane(0x0001000100010001ull << i);
    }
    chol_inv_step<14, PACKED && (CPI_SQRT_SPLIT_MASK != 0)>(a, u, acc, j);
    wave_lds_fence();   // the column reads above are complete (in-order DS) before the staging area is reused
    if (j < 15 && fl < nf) {
        if constexpr (PACKED) {
            // WITHOUT predicates: a lane stores all 15 rows of its column at its run's base; what lies beyond its diagonal (the -0
            // entries) falls into a LATER column's run, whose owner stores that slot in a later instruction -- rows descend, and
            // slot T(j) + i with i > j is T(j') + i' with j' > j, i' < i.  Nothing leaves the factor's 120 doubles (T(j) + 14 <= 119).
            // (one ds_write_b64 per row, kept apart: merged into a ds_write2_b64, rows 1 and 0 would put lane 0's trespass and lane 1's
            //  own entry -- the same slot -- into ONE instruction, whose lane order is not defined)
            double *col = sA + fl * MD + j * (j + 1) / 2;
#pragma unroll
            for (int i = 14; i >= 0; i--) { col[i] = u[i]; wave_lds_fence(); }
        } else {
#pragma unroll
            for (int i = 0; i < 15; i++) sA[fl * MD + j * 15 + i] = (i <= j) ? u[i] : 0.0;
        }
    }
    wave_lds_fence();
    {
        const int n2 = (nf * MD) >> 1;
        for (int i = lane; i < n2; i += 64) st16_nt(Rout + f0 * MD + 2 * i, sA[2 * i], sA[2 * i + 1]);
        if (((nf * MD) & 1) && lane == 0) Rout[f0 * MD + nf * MD - 1] = sA[nf * MD - 1];
    }
}

// Hessian contribution of a factor: what GTSAM's linear solver consumes after
// NoiseModelFactor::linearize (ImuFactorCPIv1.h:82 -> Gaussian::WhitenSystem -> JacobianFactor [A1 A2 | b] with
// A1 = R H1, A2 = R H2, b = -R e) when it builds a HessianFactor: the augmented information matrix
//     [A1 A2 b]^T [A1 A2 b]  =  [ G  g ]      G = A^T A (30x30),  g = A^T b,  f = b^T b
//                               [ g^T f ]
// 31x31 symmetric, written as its packed upper triangle (column-major packed, LAPACK 'U': entry (i, d), i <= d, at
// i + d (d + 1) / 2), 496 doubles per factor.  The whitened Jacobians never exist: with Lam = R^T R the matrix is
// Hc^T Lam Hc, Hc = [H1 H2 -e], and H1 / H2 are sparse in 3x3 blocks (13 of 25 and the diagonal), so rows of Z = Lam H1
// and columns of the result are short chains of 3-vector x block products (cpi_math.hpp: hsn).
// 16 lanes (one DPP row) per factor, 4 factors per wavefront.  Lane q < 15: row q of Lam (from R, broadcast reads) and of
// Z into the exchange arrays, then packed columns q and 15 + q; lane 15: column 30 by the same code applied to -y.  The
// packed triangle is written WITHOUT predicates: a lane stores all 15 entries of its runs, what lies beyond its diagonal
// falls into a later column's space and is overwritten by its owner, who stores later (hsn / tests/hostsim pin the order).
// Every product with an entry of R or of the block table is a v_fmac_f64_dpp with a row broadcast as its operand (BlkTab
// below): no LDS read and no register per entry.  17.3 KB of LDS and <= 256 registers per wavefront: two wavefronts per
// SIMD.  History: round 2 formed 31 whitened columns and 496 length-15 dot products (4 362 instructions per wavefront of 4
// factors, 1 wavefront per SIMD: 3.4 ms per 1 M factors); a first block version with 5 lanes per factor and 12 factors per
// wavefront was correct and cut the arithmetic 2.3x but sat at one wavefront per SIMD with 154 doubles of results per lane
// (2.17 ms); this mapping with the block table in LDS and two-instruction broadcasts: 1.78 ms; as shipped: 1.50 ms.
constexpr int HESS_PACKED = 496;
// Row q of Lam = R^T R with lane q of a 16-lane DPP row holding column q of R (own[k] = R[k][q], zeros below the diagonal):
// Lam[q][c] = sum_{k <= c} R[k][q] R[k][c], and R[k][c] is register own[k] of lane c -- the broadcast operand of a DPP
// multiply-add instead of 120 LDS reads per lane: the LDS pipe of a CU, shared by its eight wavefronts, is what bounds
// this kernel.  Same terms in the same order as hsn::lambda_row (the host twin).
template <int C>
__device__ __forceinline__ void lambda_row_dpp(const double (&own)[15], double (&l)[15]) {
    if constexpr (C < 15) {
        double a = dpp_mul<C>(own[0], own[0]);
#pragma unroll
        for (int k = 1; k <= C; k++) dpp_fmac<C>(a, own[k], own[k]);
        l[C] = a;
        lambda_row_dpp<C + 1>(own, l);
    }
}
// The block table of a factor (hsn: B_B .. B_DT, 106 doubles) spread over the factor's 16 lanes: entry i is register i >> 4
// of lane i & 15, so a product with a table entry is ONE double-precision DPP instruction (dpp_mul / dpp_fmac) -- no LDS
// read, no register for the entry, and the three-blocks-at-a-time loads that set the register budget of the first version
// are gone.  The table passes through LDS once (five lanes produce it, sixteen pick up their seven entries).
// Entries 0 .. 53 are the state-dependent blocks (hsn::B_B .. B_RK, row-major, written by state_blocks_column); 54 .. 90
// are the measurement's bias Jacobians and dt EXACTLY as the input record holds them (fin::O_JB .. O_DT: column-major
// blocks) -- the lanes pick those up from the record itself; 91 .. 105 is the residual.
constexpr int TAB_R = 7, TAB_D = 16 * TAB_R;
constexpr int TB_JB = 54, TB_JA = 63, TB_HB = 72, TB_HA = 81, TB_DT = 90, TB_ERR = 91, TB_END = 106;
static_assert(fin::O_JA - fin::O_JB == 9 && fin::O_HB - fin::O_JB == 18 && fin::O_HA - fin::O_JB == 27 && fin::O_DT - fin::O_JB == 36,
              "the record keeps J_beta, J_alpha, H_beta, H_alpha, dt in one run");
struct BlkTab { double r[TAB_R]; };
template <int I> __device__ __forceinline__ double tmul(const BlkTab &T, double x) { return dpp_mul<(I & 15)>(T.r[I >> 4], x); }
template <int I> __device__ __forceinline__ void tfmac(double &acc, const BlkTab &T, double x) { dpp_fmac<(I & 15)>(acc, T.r[I >> 4], x); }
// M^T v for the row-major block at table offset OFF (hsn::vTm / mulT(ldb(.), v)): same terms, same order
template <int OFF>
__device__ __forceinline__ V3 tab_mulT(const BlkTab &T, V3 v) {
    V3 o;
    o.x = tmul<OFF + 0>(T, v.x); tfmac<OFF + 3>(o.x, T, v.y); tfmac<OFF + 6>(o.x, T, v.z);
    o.y = tmul<OFF + 1>(T, v.x); tfmac<OFF + 4>(o.y, T, v.y); tfmac<OFF + 7>(o.y, T, v.z);
    o.z = tmul<OFF + 2>(T, v.x); tfmac<OFF + 5>(o.z, T, v.y); tfmac<OFF + 8>(o.z, T, v.z);
    return o;
}
template <int OFF>
__device__ __forceinline__ void tab_mulT_cm_acc(V3 &o, const BlkTab &T, V3 v) {      // the same for a COLUMN-major block
    tfmac<OFF + 0>(o.x, T, v.x); tfmac<OFF + 1>(o.x, T, v.y); tfmac<OFF + 2>(o.x, T, v.z);
    tfmac<OFF + 3>(o.y, T, v.x); tfmac<OFF + 4>(o.y, T, v.y); tfmac<OFF + 5>(o.y, T, v.z);
    tfmac<OFF + 6>(o.z, T, v.x); tfmac<OFF + 7>(o.z, T, v.y); tfmac<OFF + 8>(o.z, T, v.z);
}
template <int OFF>
__device__ __forceinline__ void tab_mulT_acc(V3 &o, const BlkTab &T, V3 v) {
    tfmac<OFF + 0>(o.x, T, v.x); tfmac<OFF + 3>(o.x, T, v.y); tfmac<OFF + 6>(o.x, T, v.z);
    tfmac<OFF + 1>(o.y, T, v.x); tfmac<OFF + 4>(o.y, T, v.y); tfmac<OFF + 7>(o.y, T, v.z);
    tfmac<OFF + 2>(o.z, T, v.x); tfmac<OFF + 5>(o.z, T, v.y); tfmac<OFF + 8>(o.z, T, v.z);
}
// g = H1^T v (hsn::h1t_vec; with v = a row of Lam: hsn::z_row), the sums of block products as running accumulations
__device__ __forceinline__ void tab_h1t(const double (&v)[15], const BlkTab &T, double (&g)[15]) {
    using namespace hsn;
    const V3 vt = ldv(v), vg = ldv(v + 3), vv = ldv(v + 6), va = ldv(v + 9), vp = ldv(v + 12);
    V3 a = tab_mulT<B_B>(T, vt); tab_mulT_acc<B_E>(a, T, vv); tab_mulT_acc<B_F>(a, T, vp);
    put3(g + 0, a);
    V3 b = mk(0.0, 0.0, 0.0); tab_mulT_cm_acc<TB_JB>(b, T, vv); tab_mulT_cm_acc<TB_JA>(b, T, vp);
    V3 c = -vg; tab_mulT_acc<B_C>(c, T, vt);
    put3(g + 3, c - b);
    V3 s = vv;                                                        // dt vp + vv
    tfmac<TB_DT>(s.x, T, vp.x); tfmac<TB_DT>(s.y, T, vp.y); tfmac<TB_DT>(s.z, T, vp.z);
    put3(g + 6, -tab_mulT<B_RK>(T, s));
    put3(g + 12, -tab_mulT<B_RK>(T, vp));
    V3 h = va; tab_mulT_cm_acc<TB_HB>(h, T, vv); tab_mulT_cm_acc<TB_HA>(h, T, vp);
    put3(g + 9, -h);
}
template <int C>
__device__ __forceinline__ void tab_nedot(double &acc, const double (&v)[15], const BlkTab &T) {   // - e . v
    if constexpr (C < 15) { dpp_fnmac<((TB_ERR + C) & 15)>(acc, T.r[(TB_ERR + C) >> 4], v[C]); tab_nedot<C + 1>(acc, v, T); }
}
// hsn::h2t_vec
__device__ __forceinline__ void tab_h2t(const double (&w)[15], const BlkTab &T, double (&t)[15]) {
    using namespace hsn;
    put3(t + 0, tab_mulT<B_A>(T, ldv(w)));
    put3(t + 3, ldv(w + 3));
    put3(t + 6, tab_mulT<B_RK>(T, ldv(w + 6)));
    put3(t + 9, ldv(w + 9));
    put3(t + 12, tab_mulT<B_RK>(T, ldv(w + 12)));
}
// TRI: R arrives as its packed upper triangle (cpi_factor_hessian_tri_batch): 480 instead of 900 doubles per wavefront to
// fetch; the lane's column of R is completed with the zeros the dense form stores below the diagonal -- same registers, same bits.
// Round 6: LDS is what capped this kernel's occupancy once the pinned core (cpi_math.hpp: CPI_CORE_PIN) had taken its registers from
// 190 / 238 to 128: the two exchange arrays (Lam, Z: 2 x 8.6 KB) and the four-factor output stage (16.4 KB) allowed 9 wavefronts per CU.
// They now TIME-SHARE one array -- Lam rows (with -y in their spare 16th column) are written, combined into the G22 / g2 runs (t), and
// only then do the Z rows take their place -- and the stage is filled and flushed two factors at a time: [records | R | tables] is
// the largest tenant, 11.1 KB with R packed (14.5 KB dense): 12 (11) wavefronts per CU at three per SIMD.  Same arithmetic in the
// same order as before (tests/hostsim's lane emulation and the whitened sweep still pin it).
// Late round 6: the table's STAGE keeps only what is not in the record already -- the 54 state-dependent entries and the residual,
// 69 doubles (pitch BLK_P) instead of 112 per factor -- which takes [records | R | tables] with R packed from 11.1 to 9.8 KB: sixteen
// wavefronts per CU, FOUR per SIMD (the registers, 128, allowed that all along); 13.1 KB and three per SIMD with R dense.  By then the
// kernel was bound by its INSTRUCTION COUNT (1 754 vector instructions per wavefront, the vector pipe 87 % busy by the counters),
// so the rest of the late changes remove instructions, not bytes: the quaternion products normalise by a reciprocal square root
// (cpi_math.hpp: CPI_QUAT_RECIP; -250), lane 15 gets w = -y out of the other lanes' row combination (below; -45 and 8 LDS reads),
// -y and f come out of negated multiply-adds (dpp_fnmac), the triangle's rows are read unclamped, the output slots and the flush are
// instruction offsets from pointers formed once (-40): 1 516 per wavefront, 1.30 -> 1.20 ms per 1 M factors with R packed, 1.42 ->
// 1.27 dense (profiles/r06_small_sweeps.md section 5).
#ifndef CPI_HESS_WPS
#define CPI_HESS_WPS 3
#endif
#ifndef CPI_HESS_WPS_TRI
#define CPI_HESS_WPS_TRI 4
#endif
#ifndef CPI_HESS_CORE_LANES
#define CPI_HESS_CORE_LANES 5
#endif
constexpr int TBK_ERR = TB_JB, BLK_P = 70;      // where the residual sits in the stage; the stage's pitch (even: 16-byte rows)
static_assert(TBK_ERR + (TB_END - TB_ERR) <= BLK_P, "the stage holds the state-dependent blocks and the residual");
template <int MODEL, bool TRI = false>
__global__ __launch_bounds__(64, TRI ? CPI_HESS_WPS_TRI : CPI_HESS_WPS) void cpi_factor_hessian_kernel(FactorArgs A, double *hess) {
    using namespace hsn;
    constexpr int FPW = 4, IN_D = fin::IN_D, RD = TRI ? CPI_TRI_DOUBLES : 225;
    // [input records | R | block tables] -> ONE exchange array (Lam, then Z) -> the output stage of two factors (+ 128 trash slots)
    // the exchange array of a factor: 15 rows of Lam / Z pitched ROWP, -y in column 15 AND as a sixteenth row (see the column phase)
    constexpr int XM = 16 * ROWP + 14;     // 302 doubles = 28 dwords mod 64 apart, as the 270 of the 15-row array were
    constexpr int U1 = FPW * XM, HEAD = FPW * IN_D + FPW * RD + FPW * BLK_P, STAGE = 2 * HESS_PACKED + 128;
    constexpr int LDS_D = (HEAD > U1 ? HEAD : U1) > STAGE ? (HEAD > U1 ? HEAD : U1) : STAGE;
    static_assert(TB_END <= TAB_D && XM >= 16 * ROWP && ROWP >= 16 && B_RK + 9 <= TBK_ERR, "table / exchange geometry");
    __shared__ __attribute__((aligned(16))) double sAll[LDS_D];
    __shared__ double sDummy[2];
    double *sU1 = sAll;
    double *sR = sAll + FPW * IN_D, *sTab = sAll + FPW * IN_D + FPW * RD;
    const int lane = threadIdx.x;
    const long long grp = factor_group_of_block((A.F + FPW - 1) / FPW);
    if (grp < 0) return;
    const long long f0 = grp * FPW;
    const int nf = (int)min((long long)FPW, A.F - f0);
    const int q = lane & 15, f = min(lane >> 4, nf - 1);     // missing factors shadow the last one (same values, same slots)
    const int qr = min(q, 14);                               // lane 15 shadows row 14 in the row phase

    // ---- the input records
    constexpr int RT = (FPW * RD + 63) / 64;
    {
        // the R matrices of the wavefront (900 doubles, or 480 packed; coalesced) travel with the input records: ONE memory round trip
        double rr[RT];
#pragma unroll
        for (int t = 0; t < RT; t++) rr[t] = A.sqrt_info[f0 * RD + min(lane + 64 * t, nf * RD - 1)];
        factor_fetch_inputs<MODEL, FPW, false>(A, f0, nf, lane, sU1, sDummy);
#pragma unroll
        for (int t = 0; t < RT; t++)
            if (lane + 64 * t < FPW * RD) sR[lane + 64 * t] = rr[t];
    }
    __syncthreads();

    // ---- block table of the factor: the state-dependent blocks (three column tasks), the measurement's bias Jacobians,
    // the residual -- through LDS into the lanes' registers
    BlkTab T;
    V3 dcol;                                                 // the lane's column of its diagonal block of H2
    {
        double *blk = sTab + f * BLK_P;
        if (q < CPI_HESS_CORE_LANES) {
            // (only lanes 0-2 and 4 of a factor use what the core computes: the others sit it out -- the same instructions for the
            //  wavefront, a quarter of the FP64 lanes switching during a third of the kernel; CPI_HESS_CORE_LANES = 16: everybody)
            const FactorMeas m = factor_meas_of(sU1 + f * IN_D, A.grav);
            FactorShared S;
            V3 e5[5];
            factor_shared_core<MODEL>(m, S, e5);
            if (q < 3) state_blocks_column<MODEL>(S, m, q, blk);
            else if (q == 4) {
#pragma unroll
                for (int a = 0; a < 5; a++) { blk[TBK_ERR + 3 * a] = e5[a].x; blk[TBK_ERR + 3 * a + 1] = e5[a].y; blk[TBK_ERR + 3 * a + 2] = e5[a].z; }
            }
        }
        wave_lds_fence();
        const double *rec = sU1 + f * IN_D + (fin::O_JB - TB_JB);
#pragma unroll
        for (int r = 0; r < TAB_R; r++) {                              // entries past TB_END: never used
            const int i = r * 16 + q;
            const bool from_record = (r * 16 + 15 >= TB_JB) && (r * 16 < TB_ERR) && i >= TB_JB && i < TB_ERR;
            const int bi = (i < TB_JB) ? i : min(i, TB_END - 1) - (TB_ERR - TBK_ERR);      // the stage skips what the record holds
            T.r[r] = *(from_record ? rec + i : blk + bi);
        }
        dcol = h2_diag_col(blk, (q < 15) ? q / 3 : 1, (q < 15) ? q % 3 : 0);      // lane 15: a column of an identity block, (1, 0, 0)
    }

    // ---- row phase: row q of Lam and of Z, y_q -- in registers; Lam (+ y in column 15) goes to the exchange array first
    double *xch = sAll + f * XM;
    double z[15];
    {
        double l[15], own[15];
#pragma unroll
        for (int k = 0; k < 15; k++) {
            // (rows past the diagonal read on into the next columns of the same triangle -- T(qr) + 14 <= 119 -- and are replaced by the zeros)
            if constexpr (TRI) { const double v = sR[f * RD + qr * (qr + 1) / 2 + k]; own[k] = (k <= qr) ? v : 0.0; }
            else own[k] = sR[f * 225 + qr * 15 + k];
        }
        // the selects above are VALU writes and lambda_row_dpp reads own[] as DPP sources right away: all of them complete, then two
        // wait states (the hazard recogniser does not see into the inline assembly of dpp_fmac; tests/tools/dpp_hazards.py does)
        asm volatile("s_nop 1" : "+v"(own[0]), "+v"(own[1]), "+v"(own[2]), "+v"(own[3]), "+v"(own[4]), "+v"(own[5]), "+v"(own[6]), "+v"(own[7]),
                     "+v"(own[8]), "+v"(own[9]), "+v"(own[10]), "+v"(own[11]), "+v"(own[12]), "+v"(own[13]), "+v"(own[14]));
        lambda_row_dpp<0>(own, l);
        tab_h1t(l, T, z);
        double ny = 0.0;      // -y_q = -(Lam e)_q: the running sum takes its terms negated (source modifier of the DPP multiply-add)
        tab_nedot<0>(ny, l, T);
        wave_lds_fence();     // every lane has read R (and its table entries) before the area becomes the exchange array
#pragma unroll
        for (int c = 0; c < 15; c++) xch[qr * ROWP + c] = l[c];
        // -y twice: as column 15 (what lane 15 reads as "its column of Z" in part 2 -- the Z rows leave it alone) and as row 15, which
        // lane 15 combines in part 1 with the identity column (1, 0, 0) as its block column: w = -y by the code of the other lanes,
        // where fifteen reads, negations and selects stood
        xch[qr * ROWP + 15] = ny;
        xch[15 * ROWP + qr] = ny;
    }
    wave_lds_fence();

    // ---- column phase, part 1 (needs Lam): the G22 / g2 run t.  hsn::lane_columns is the host twin of the whole phase.
    // rows 3 j .. 3 j + 2 of the array for a lane q < 15; lane 15: row 15 three times (the other two meet the zeros of its dcol)
    const double *xr0 = xch + ((q < 15) ? 3 * (q / 3) : 15) * ROWP;
    const int xrs = (q < 15) ? ROWP : 0;
    double g[15], u[15], t[15], fq;
    {
        double w[15];
        rows_comb3(xr0, xr0 + xrs, xr0 + 2 * xrs, dcol, w);
        tab_h2t(w, T, t);
    }
    wave_lds_fence();         // Lam is consumed: the Z rows take its place (column 15 keeps -y)
#pragma unroll
    for (int c = 0; c < 15; c++) xch[qr * ROWP + c] = z[c];
    wave_lds_fence();
    // ---- part 2 (needs Z): the G11 / g1 / f run g, fq and the G12 run u
    {
        double zc[15];
#pragma unroll
        for (int k = 0; k < 15; k++) zc[k] = xch[k * ROWP + q];             // column q of Z; for q = 15: -y
        tab_h1t(zc, T, g);
        fq = 0.0;
        tab_nedot<0>(fq, zc, T);                                            // lane 15: f = -(e . -y)
        rows_comb3(xr0, xr0 + xrs, xr0 + 2 * xrs, dcol, u);
    }
    wave_lds_fence();

    // ---- out through the stage, two factors at a time: their packed triangles are one contiguous 7.9 KB run of the output.
    // (late round 6: the lane's slots are formed once, every store and every piece of the flush is an instruction offset from them --
    //  the runs that have no place in the triangle, lane 15's u and the other lanes' f, go to trash slots of their own instead of
    //  through a select per element; a full pair of factors is flushed by eight unrolled pieces per lane, the last one by 48 lanes)
    {
        double *st = sAll + (f & 1) * HESS_PACKED, *trash = sAll + 2 * HESS_PACKED;
        double *pg = st + ((q < 15) ? pk(0, q) : pk(0, 30));
        double *pt = st + ((q < 15) ? pk(15, 15 + q) : pk(15, 30));
        double *pu = (q < 15) ? st + pk(0, 15 + q) : trash + (lane >> 4) * 15;
        double *pf = (q == 15) ? st + pk(30, 30) : trash + 64 + lane;
        const double *piece = sAll + 2 * lane;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const int cnt = min(2, nf - 2 * pass);               // wave-uniform
            if (cnt <= 0) break;
            if ((f >> 1) == pass) {
#pragma unroll
                for (int r = 14; r >= 0; r--) pg[r] = g[r];        // descending: the owner of an entry stores it after every trespasser
                wave_lds_fence();
#pragma unroll
                for (int r = 0; r < 15; r++) pt[r] = t[r];
                wave_lds_fence();
#pragma unroll
                for (int r = 0; r < 15; r++) pu[r] = u[r];
                *pf = fq;
            }
            wave_lds_fence();
            double *dst = hess + (f0 + 2 * pass) * HESS_PACKED + 2 * lane;
            if (cnt == 2) {
#pragma unroll
                for (int it = 0; it < 8; it++)
                    if (it < 7 || lane < HESS_PACKED - 7 * 64) st16_nt(dst + 128 * it, piece[128 * it], piece[128 * it + 1]);
            } else {                                             // the odd factor at the end of a launch
                for (int idx = lane; idx < HESS_PACKED / 2; idx += 64) st16_nt(dst + 2 * (idx - lane), sAll[2 * idx], sAll[2 * idx + 1]);
            }
            wave_lds_fence();     // the flush has read the stage (in-order DS) before the next pair overwrites it
        }
    }
}

// getpredictedstate_v1 / _v2 (GraphSolver_IMU.cpp:263-307): 344 bytes per factor in and out, ~130 FP64 instructions -- a copy with a
// little arithmetic in it.  One wavefront per 64 factors, one lane per factor for the arithmetic, but NOT for the memory traffic:
// until round 5 every lane fetched its own 128-byte state, its 24 / 24 / 32-byte measurement fields and stored its 128-byte result
// with 8- and 16-byte accesses strided by the record size (0.072 / 0.089 ms per 1 M factors = 0.60 / 0.48 of 8 TB/s, where a plain
// copy of the same mix reaches 0.88: tools/exp/mix_probe.hip, profiles/r06_small_sweeps.md).  Now the wavefront moves its 64 records
// as ONE burst of fully coalesced pieces -- the states (gathered through idx_i or chained) as 16-byte pieces, eight lanes per
// state; the SoA measurement fields as consecutive doubles -- all requested before the first is used, parked in LDS record-major
// (pitch 27 doubles: odd, so the per-lane record reads hit 32 distinct banks), and the 64 results leave through the same area as 512
// consecutive 16-byte non-temporal stores.
constexpr int PRED_IN_D = 27, PRED_OUT_D = 17;   // LDS pitches (doubles): state 16 + alpha 3 + beta 3 + q 4 + DT 1; result 16 (+1: odd)
template <int MODEL>
__global__ __launch_bounds__(64) void cpi_predict_kernel(PredictArgs A) {
    constexpr int FPW = 64;
    __shared__ __attribute__((aligned(16))) double sRec[FPW * PRED_IN_D];
    static_assert(FPW * PRED_OUT_D <= FPW * PRED_IN_D, "the results re-use the record area");
    const int lane = threadIdx.x;
    const long long f0 = (long long)blockIdx.x * FPW;
    const int nf = (int)min((long long)FPW, A.F - f0);
    struct __attribute__((packed, aligned(8))) d2u { double a, b; };
    // ---- one burst: 8 x 16-byte state pieces per lane (piece p = lane + 64 r: part p & 7 of the state of factor p >> 3), then the fields
    d2u st[8];
    {
        long long si[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const long long ff = f0 + min((lane >> 3) + 8 * r, nf - 1);
            // branch-free NULL handling (a valid dummy address is read and discarded) keeps all loads in one block
            const int vi = (A.idx_i ? A.idx_i : reinterpret_cast<const int *>(A.states_i))[ff];
            si[r] = min(max(A.idx_i ? (long long)vi : ff, 0ll), A.S - 1);
        }
#pragma unroll
        for (int r = 0; r < 8; r++) st[r] = *reinterpret_cast<const d2u *>(A.states_i + si[r] * 16 + 2 * (lane & 7));
    }
    FieldFetch<FPW, 3> f_alpha, f_beta; FieldFetch<FPW, 4> f_q; FieldFetch<FPW, 1> f_dt;
    f_alpha.load(A.meas.alpha, f0, nf, lane); f_beta.load(A.meas.beta, f0, nf, lane); f_q.load(A.meas.q, f0, nf, lane);
    f_dt.load(A.meas.DT, f0, nf, lane);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        double *d = sRec + ((lane >> 3) + 8 * r) * PRED_IN_D + 2 * (lane & 7);
        d[0] = st[r].a; d[1] = st[r].b;
    }
    f_alpha.store(sRec, PRED_IN_D, 16, lane); f_beta.store(sRec, PRED_IN_D, 19, lane); f_q.store(sRec, PRED_IN_D, 22, lane);
    f_dt.store(sRec, PRED_IN_D, 26, lane);
    wave_lds_fence();
    // ---- lane = factor (lanes past the last factor redo it: same values, and they do not store)
    NavState o;
    {
        const double *rec = sRec + min(lane, nf - 1) * PRED_IN_D;
        const NavState xi = ld_state(rec);
        o = predict_state<MODEL>(xi, ldv3(rec + 16), ldv3(rec + 19), ldq4(rec + 22), rec[26], mk(A.grav[0], A.grav[1], A.grav[2]));
    }
    wave_lds_fence();   // every record is read (in-order DS) before the area becomes the output stage
    {
        double *d = sRec + lane * PRED_OUT_D;
        d[0] = o.q.x; d[1] = o.q.y; d[2] = o.q.z; d[3] = o.q.w;
        stv3(d + 4, o.bg); stv3(d + 7, o.v); stv3(d + 10, o.ba); stv3(d + 13, o.p);
    }
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int fl = (lane >> 3) + 8 * r;
        const double *src = sRec + fl * PRED_OUT_D + 2 * (lane & 7);
        if (fl < nf) st16_nt(A.states_j + (f0 + fl) * 16 + 2 * (lane & 7), src[0], src[1]);
    }
}


}  // namespace
