// hostsim.cpp -- lane-by-lane HOST emulation of the HIP kernels' arithmetic (cpi_math.hpp).
// TEST INFRASTRUCTURE ONLY: lets the CPU test-suite validate the kernel mathematics (segment
// composition, column-lane covariance recursion with its transpose exchange, factor blocks)
// against the oracle where no GPU exists.  Not linked into, or reachable from, libcpi_amd.so.
#include "../../cpi_amd/csrc/cpi_math.hpp"
#include <algorithm>
#include <cstring>
#include <array>
#include <vector>
using namespace cpi;

namespace {
const int OUTD = 308;  // DT1 alpha3 beta3 q4 R9 Jq9 Ja9 Jb9 Ha9 Hb9 Oa9 Ob9 P225
void put_cm(double *dst, const M3 &A) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) dst[j * 3 + i] = A.m[i][j]; }
V3 ld3(const double *p) { return mk(p[0], p[1], p[2]); }
M3 ld_cm(const double *p) { M3 A; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A.m[i][j] = p[j * 3 + i]; return A; }

template <int MODEL, bool JAC, bool AVG>
void mean_window(int L, int n, const double *kn, const double *lin, const double *qk, const double *grav, double *o) {
    const V3 bw = ld3(lin), ba = ld3(lin + 3);
    V3 gk = mk(0, 0, 0);
    if (MODEL == 2) gk = mul(quat_2_Rot(ldq(qk)), ld3(grav));
    const int per = (n + L - 1) / L;
    if constexpr (MODEL == 2 && !JAC) {
        if (L > 1) {   // several lanes per window: segment form with the gravity response matrices (as the kernel)
            std::vector<MeanState<false>> sg(L);
            std::vector<GravAcc> ga(L);
            for (int l = 0; l < L; l++) {
                mean_init(sg[l]); grav_init(ga[l]);
                const int s0 = std::min(n, l * per), s1 = std::min(n, s0 + per);
                for (int s = s0; s < s1; s++) {
                    const double *k0 = kn + 7 * s, *k1 = kn + 7 * (s + 1);
                    mean_step_v2seg<AVG>(sg[l], ga[l], k0[0], k1[0], ld3(k0 + 1), ld3(k0 + 4), ld3(k1 + 1), ld3(k1 + 4), bw, ba);
                }
            }
            for (int st = 1; st < L; st *= 2)
                for (int l = 0; l + st < L; l += 2 * st) { grav_combine(ga[l], sg[l], ga[l + st], sg[l + st]); mean_combine(sg[l], sg[l + st]); }
            grav_apply(sg[0], ga[0], gk);
            const MeanState<false> &s = sg[0];
            o[0] = s.DT;
            o[1] = s.alpha.x; o[2] = s.alpha.y; o[3] = s.alpha.z;
            o[4] = s.beta.x; o[5] = s.beta.y; o[6] = s.beta.z;
            const Q4 q = rot_2_quat(s.R);
            o[7] = q.x; o[8] = q.y; o[9] = q.z; o[10] = q.w;
            put_cm(o + 11, s.R);
            return;
        }
    }
    std::vector<MeanState<JAC>> seg(L);
    for (int l = 0; l < L; l++) {
        mean_init(seg[l]);
        const int s0 = std::min(n, l * per), s1 = std::min(n, s0 + per);
        for (int s = s0; s < s1; s++) {
            const double *k0 = kn + 7 * s, *k1 = kn + 7 * (s + 1);
            mean_step<MODEL, JAC, AVG>(seg[l], k0[0], k1[0], ld3(k0 + 1), ld3(k0 + 4), ld3(k1 + 1), ld3(k1 + 4), bw, ba, gk);
        }
    }
    // order-preserving tree (as the shuffle tree in the kernel): stride doubles
    for (int st = 1; st < L; st *= 2)
        for (int l = 0; l + st < L; l += 2 * st) mean_combine(seg[l], seg[l + st]);
    const MeanState<JAC> &s = seg[0];
    o[0] = s.DT;
    o[1] = s.alpha.x; o[2] = s.alpha.y; o[3] = s.alpha.z;
    o[4] = s.beta.x; o[5] = s.beta.y; o[6] = s.beta.z;
    const Q4 q = rot_2_quat(s.R);
    o[7] = q.x; o[8] = q.y; o[9] = q.z; o[10] = q.w;
    put_cm(o + 11, s.R);
    if (JAC) {
        put_cm(o + 20, s.Jq); put_cm(o + 29, s.Ja); put_cm(o + 38, s.Jb); put_cm(o + 47, s.Ha); put_cm(o + 56, s.Hb);
        put_cm(o + 65, s.Oa); put_cm(o + 74, s.Ob);
    }
}

template <int MODEL, bool AVG>
void cov_window(int n, const double *kn, const double *lin, const double *qk, const double *sig, const double *grav, double *o) {
    typedef CovDims<MODEL> D;
    const V3 bw = ld3(lin), ba = ld3(lin + 3);
    V3 gk = mk(0, 0, 0);
    if (MODEL == 2) gk = mul(quat_2_Rot(ldq(qk)), ld3(grav));
    const double q4[4] = { sig[0] * sig[0], sig[1] * sig[1], sig[2] * sig[2], sig[3] * sig[3] };
    const int NL = D::GROUP;   // all lanes of the group, idle ones included (as in the kernel)
    const int CH = D::GROUP;   // intervals per phase-A pass
    std::vector<CovLane<MODEL>> lane(NL);
    const int IRD = IrPitch<MODEL>::V;
    std::vector<double> exch(exch_doubles(1), 0.0), irs(CH * IRD, 0.0);
    double gs[GS_DOUBLES];
    cov_gs_init(gs);
    std::vector<int> colof(NL);
    for (int j = 0; j < NL; j++) { colof[j] = cov_col_of_lane<MODEL>(j); cov_init(lane[j], colof[j], q4); cov_exch_init<MODEL>(exch.data(), 1, colof[j], q4); }
    for (int base = 0; base < n; base += CH) {
        // ---- phase A, as the kernel does it: per-lane closed forms, Hillis-Steele prefix product of the step
        // rotations, finish_interval, ordered tree reduction of the mean increments
        std::vector<SampleRec> r(CH);
        std::vector<M3> inc(CH);
        for (int sl = 0; sl < CH; sl++) {
            const int s = base + sl;
            if (s < n) {
                const double *k0 = kn + 7 * s, *k1 = kn + 7 * (s + 1);
                r[sl] = make_sample_rec<MODEL, AVG>(k0[0], k1[0], ld3(k0 + 1), ld3(k0 + 4), ld3(k1 + 1), ld3(k1 + 4), bw, ba);
            } else {
                r[sl].dt = 0; r[sl].w = mk(0, 0, 0); r[sl].a0 = mk(0, 0, 0); r[sl].a1 = mk(0, 0, 0);
                r[sl].f1 = r[sl].f2 = r[sl].f3 = r[sl].f4 = 0; r[sl].Rstep = eye(); r[sl].Rhalf = eye();
            }
            inc[sl] = r[sl].Rstep;
        }
        for (int d = 1; d < CH; d <<= 1) {
            std::vector<M3> prev = inc;
            for (int sl = d; sl < CH; sl++) inc[sl] = mm(prev[sl], prev[sl - d]);
        }
        const M3 Rc = rec_mat(gs, GS_R);
        std::vector<MeanInc> mi(CH);
        for (int sl = 0; sl < CH; sl++) {
            const M3 pre = (sl == 0) ? eye() : inc[sl - 1];
            mi[sl] = finish_interval<MODEL, AVG>(r[sl], mm(pre, Rc), gk, irs.data() + sl * IRD);
        }
        for (int d = 1; d < CH; d <<= 1)
            for (int sl = 0; sl + d < CH; sl += 2 * d) mi[sl] = inc_combine(mi[sl], mi[sl + d]);
        gs_apply_inc(gs, mi[0]);
        rec_put_mat(gs, GS_R, mm(inc[CH - 1], Rc));
        // ---- phase C
        const int cnt = std::min(CH, n - base);
        M3 Rs = Rc;   // stage-0 rotation: the pass-start rotation, then each interval's R_new (as the kernel carries it)
        for (int sl = 0; sl < cnt; sl++) {
            const double *ir = irs.data() + sl * IRD;
            for (int j = 0; j < NL; j++) cov_begin<MODEL>(lane[j], ir, cov_h_offset<MODEL>(colof[j]));
            for (int st = 0; st < 4; st++) {
                double M[32][9];
                for (int j = 0; j < NL; j++) {
                    cov_stage_M(lane[j], st, (st == 0) ? Rs : cov_stage_rotation<MODEL>(ir, st), M[j]);
                    if (colof[j] < D::NPCOL)
                        for (int rr = 0; rr < CovExchRows<MODEL>::V; rr++) exch[rr * EXCH_PITCH + exch_pos<MODEL>(colof[j])] = M[j][rr];
                }
                if (CovPBySymmetry<MODEL>::V) {
                    // the kernel's masked row_shr:6 DPP move: lanes 12..15 take the stage's X of lanes 6..9 as their row
                    double mt[32][D::NR];
                    for (int j = 0; j < NL; j++) {
                        const double *row = exch.data() + cov_row_off<MODEL>(1, 0, colof[j]);
                        for (int i = 0; i < D::NR; i++) mt[j][i] = row[exch_pos<MODEL>(i)];
                    }
                    for (int j = CovPLanes<MODEL>::FIRST; j < CovPLanes<MODEL>::FIRST + 4 && j < NL; j++)
                        for (int i = 0; i < D::NR; i++) mt[j][i] = cov_stage_X(lane[j - CovPLanes<MODEL>::SHIFT], st)[i];
                    for (int j = 0; j < NL; j++) cov_stage_finish_regs(lane[j], st, M[j], mt[j]);
                } else {
                    for (int j = 0; j < NL; j++)
                        cov_stage_finish(lane[j], st, M[j], exch.data() + cov_row_off<MODEL>(1, 0, colof[j]));
                }
            }
            Rs = cov_stage_rotation<MODEL>(ir, 3);
            for (int j = 0; j < NL; j++) cov_end(lane[j]);
            if (MODEL == 2)   // the masked row_shr:4 DPP move: lanes 4..7 <- lanes 0..3
                for (int b = 0; b < 4; b++) for (int i = 0; i < D::NR; i++) lane[4 + b].P0[i] = lane[b].P0[i];
        }
    }
    o[0] = gs[GS_DT];
    o[1] = gs[GS_ALPHA]; o[2] = gs[GS_ALPHA + 1]; o[3] = gs[GS_ALPHA + 2];
    o[4] = gs[GS_BETA]; o[5] = gs[GS_BETA + 1]; o[6] = gs[GS_BETA + 2];
    const M3 Rfin = rec_mat(gs, GS_R);
    const Q4 q = rot_2_quat(Rfin);
    o[7] = q.x; o[8] = q.y; o[9] = q.z; o[10] = q.w;
    put_cm(o + 11, Rfin);
    std::vector<int> laneof(D::NCOL + 1, 0);
    for (int j = 0; j < NL; j++) if (colof[j] < D::NCOL) laneof[colof[j]] = j;
    for (int c = 0; c < 15; c++) for (int i = 0; i < 15; i++) o[83 + c * 15 + i] = lane[laneof[c]].P0[i];
    if (MODEL == 2) {
        for (int c = 0; c < 3; c++) {
            const CovLane<MODEL> &g = lane[laneof[D::NPCOL + c]], &a = lane[laneof[D::NPCOL + 3 + c]], &l = lane[laneof[D::NPCOL + 6 + c]];
            for (int i = 0; i < 3; i++) {
                o[20 + c * 3 + i] = -g.P0[0 + i];   // J_q = -D(0:3, 3:6)
                o[29 + c * 3 + i] = g.P0[12 + i];   // J_a = D(12:15, 3:6)
                o[38 + c * 3 + i] = g.P0[6 + i];    // J_b = D(6:9, 3:6)
                o[47 + c * 3 + i] = a.P0[12 + i];   // H_a = D(12:15, 9:12)
                o[56 + c * 3 + i] = a.P0[6 + i];    // H_b = D(6:9, 9:12)
                o[65 + c * 3 + i] = l.P0[12 + i];   // O_a = D(12:15, 18:21)
                o[74 + c * 3 + i] = l.P0[6 + i];    // O_b = D(6:9, 18:21)
            }
        }
    }
}

// Forster comparator kernel, lane by lane: 16 lanes per window -- lane j < 15 owns covariance column j, lanes 0-2
// also a gyro-bias Jacobian column, lanes 3-5 an accelerometer-bias one; every lane carries the means.  The 9-row
// exchange hands each theta / v / p column the matching ROW of F P (= its column of P F^T).
void forster_window(int n, const double *kn, const double *lin, const double *sig, double *o) {
    const V3 bg = ld3(lin), ba = ld3(lin + 3);
    const double q4[4] = { sig[0] * sig[0], sig[1] * sig[1], sig[2] * sig[2], sig[3] * sig[3] };
    const int NL = 16, CH = 12, ROWS = 15;
    std::vector<double> irs(CH * fsd::IR_SIZE, 0.0), exch(ROWS * EXCH_PITCH, 0.0);
    struct LaneS { fsd::Mean m; fsd::JacCol J; double x[15]; };
    std::vector<LaneS> lane(NL);
    for (int j = 0; j < NL; j++) { fsd::mean_init(lane[j].m); fsd::jac_init(lane[j].J); for (int i = 0; i < 15; i++) lane[j].x[i] = 0.0; }
    for (int base = 0; base < n; base += CH) {
        for (int sl = 0; sl < CH; sl++) {   // phase A: one lane per interval, no dependence between them
            const int s = base + sl;
            fsd::Rec r;
            if (s < n) {
                const double *k0 = kn + 7 * s, *k1 = kn + 7 * (s + 1);
                r = fsd::make_rec(k0[0], k1[0], ld3(k0 + 1), ld3(k0 + 4), bg, ba, q4[0]);
            } else r = fsd::make_rec(0, 0, mk(0, 0, 0), mk(0, 0, 0), bg, ba, q4[0]);
            fsd::put_rec(irs.data() + sl * fsd::IR_SIZE, r);
        }
        const int cnt = std::min(CH, n - base);
        for (int sl = 0; sl < cnt; sl++) {
            const double *ir = irs.data() + sl * fsd::IR_SIZE;
            const fsd::Rec r = fsd::get_rec(ir);
            for (int j = 0; j < NL; j++) {
                LaneS &Ls = lane[j];
                const V3 eg = (j < 3) ? unit(j) : mk(0, 0, 0), ek = (j >= 3 && j < 6) ? unit(j - 3) : mk(0, 0, 0);
                fsd::jac_step(Ls.J, Ls.m.R, r, ek, eg);
                fsd::mean_step(Ls.m, r);
                double y[15];
                fsd::F_apply(r, Ls.x, y);
                if (j < 15) for (int i = 0; i < 15; i++) exch[i * EXCH_PITCH + j] = y[i];
            }
            for (int j = 0; j < NL; j++) {
                LaneS &Ls = lane[j];
                double z[15];
                for (int i = 0; i < 15; i++) z[i] = exch[(j < 15 ? j : 0) * EXCH_PITCH + i];
                fsd::F_apply(r, z, Ls.x);
                fsd::theta_noise_add(Ls.x, r, rec_v3(ir, fsd::IR_JD + 3 * std::min(j, 2)), j < 3 ? 1.0 : 0.0);
                const V3 nbg = (j >= 3 && j < 6) ? q4[1] * unit(j - 3) : mk(0, 0, 0);
                const V3 nv = (j >= 6 && j < 9) ? q4[2] * unit(j - 6) : mk(0, 0, 0);
                const V3 nba = (j >= 9 && j < 12) ? q4[3] * unit(j - 9) : mk(0, 0, 0);
                Ls.x[3] += r.dt * nbg.x; Ls.x[4] += r.dt * nbg.y; Ls.x[5] += r.dt * nbg.z;
                Ls.x[6] += r.dt * nv.x; Ls.x[7] += r.dt * nv.y; Ls.x[8] += r.dt * nv.z;
                Ls.x[9] += r.dt * nba.x; Ls.x[10] += r.dt * nba.y; Ls.x[11] += r.dt * nba.z;
            }
        }
    }
    const fsd::Mean &m = lane[0].m;
    o[0] = m.dT;
    o[1] = m.p.x; o[2] = m.p.y; o[3] = m.p.z;
    o[4] = m.v.x; o[5] = m.v.y; o[6] = m.v.z;
    M3 Rt;
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) Rt.m[i][k] = m.R.m[k][i];
    const Q4 q = rot_2_quat(Rt);
    o[7] = q.x; o[8] = q.y; o[9] = q.z; o[10] = q.w;
    put_cm(o + 11, Rt);
    for (int c = 0; c < 3; c++) {
        const fsd::JacCol &g = lane[c].J, &a = lane[3 + c].J;
        const double jq[3] = { -g.r.x, -g.r.y, -g.r.z }, ja[3] = { g.p.x, g.p.y, g.p.z }, jb[3] = { g.v.x, g.v.y, g.v.z };
        const double ha[3] = { a.p.x, a.p.y, a.p.z }, hb[3] = { a.v.x, a.v.y, a.v.z };
        for (int i = 0; i < 3; i++) {
            o[20 + c * 3 + i] = jq[i]; o[29 + c * 3 + i] = ja[i]; o[38 + c * 3 + i] = jb[i];
            o[47 + c * 3 + i] = ha[i]; o[56 + c * 3 + i] = hb[i];
        }
    }
    for (int c = 0; c < 15; c++) for (int i = 0; i < 15; i++) o[83 + c * 15 + i] = lane[c].x[i];
}
}  // namespace

extern "C" void hs_mean(int model, int jac, int avg, int L, long W, int n, const double *kn, const double *lin,
                        const double *qk, const double *grav, double *out) {
    for (long w = 0; w < W; w++) {
        const double *k = kn + (size_t)w * (n + 1) * 7, *l = lin + w * 6, *q = qk ? qk + w * 4 : nullptr;
        double *o = out + w * OUTD;
#define GO(M, J, A) mean_window<M, J, A>(L, n, k, l, q, grav, o)
        if (model == 1) { if (jac) { if (avg) GO(1, true, true); else GO(1, true, false); } else { if (avg) GO(1, false, true); else GO(1, false, false); } }
        else            { if (jac) { if (avg) GO(2, true, true); else GO(2, true, false); } else { if (avg) GO(2, false, true); else GO(2, false, false); } }
#undef GO
    }
}
extern "C" void hs_cov(int model, int avg, long W, int n, const double *kn, const double *lin, const double *qk,
                       const double *sig, const double *grav, double *out) {
    for (long w = 0; w < W; w++) {
        const double *k = kn + (size_t)w * (n + 1) * 7, *l = lin + w * 6, *q = qk ? qk + w * 4 : nullptr;
        double *o = out + w * OUTD;
        if (model == 1) { if (avg) cov_window<1, true>(n, k, l, q, sig, grav, o); else cov_window<1, false>(n, k, l, q, sig, grav, o); }
        else            { if (avg) cov_window<2, true>(n, k, l, q, sig, grav, o); else cov_window<2, false>(n, k, l, q, sig, grav, o); }
    }
}
// rec: 87 doubles in the order of oracle_py.FACTOR_FIELDS
extern "C" void hs_factor(int model, long F, const double *rec, const double *xi, const double *xj, double *err,
                          double *H1, double *H2) {
    for (long k = 0; k < F; k++) {
        const double *r = rec + k * 87;
        FactorMeas f;
        f.alpha = r; f.beta = r + 3; f.q_KtoK1 = r + 6;
        double lin6[6] = { r[13], r[14], r[15], r[10], r[11], r[12] };   // record order is ba_lin, bg_lin
        f.lin = lin6;
        f.J_q = r + 16; f.J_beta = r + 25; f.J_alpha = r + 34; f.H_beta = r + 43; f.H_alpha = r + 52;
        f.dt = r + 61; f.grav = ld3(r + 62); f.q_K_lin = r + 65; f.O_beta = r + 69; f.O_alpha = r + 78;
        f.xi = xi + k * 16; f.xj = xj + k * 16;
        for (int c = 0; c < 15; c++) {
            if (model == 1) factor_eval_col<1>(f, c, err[k * 15 + c], H1 + k * 225 + c * 15, H2 + k * 225 + c * 15);
            else factor_eval_col<2>(f, c, err[k * 15 + c], H1 + k * 225 + c * 15, H2 + k * 225 + c * 15);
        }
    }
}
// Hessian blocks by 3x3 block algebra (cpi_math.hpp: hsn), the sixteen lanes of a factor run one after the other exactly
// as cpi_factor_hessian_kernel runs them side by side: block table from the shared algebra, rows of Lam and Z into the
// exchange arrays, then a packed column (or two) per lane, stored with the kernel's own "no predicate" order.
// R [F][225] column-major upper triangular; out [F][496] packed upper triangle.
template <int MODEL>
static void hessian_factor(const FactorMeas &f, const double *R, double *out) {
    using namespace hsn;
    double blk[BLK_D] = {0}, lam[MAT_D], zx[MAT_D];
    FactorShared S;
    V3 e5[5];
    factor_shared_core<MODEL>(f, S, e5);
    for (int cc = 0; cc < 3; cc++) state_blocks_column<MODEL>(S, f, cc, blk);
    stb(blk + B_JB, ldcm(f.J_beta)); stb(blk + B_JA, ldcm(f.J_alpha)); stb(blk + B_HB, ldcm(f.H_beta)); stb(blk + B_HA, ldcm(f.H_alpha));
    for (int a = 0; a < 5; a++) { blk[B_ERR + 3 * a] = e5[a].x; blk[B_ERR + 3 * a + 1] = e5[a].y; blk[B_ERR + 3 * a + 2] = e5[a].z; }
    blk[B_DT] = f.dt[0];
    for (int q = 0; q < 15; q++) {
        double l[15], z[15], y;
        lambda_row(R, q, l);
        z_row(l, blk, z, y);
        for (int c = 0; c < 15; c++) { lam[q * ROWP + c] = l[c]; zx[q * ROWP + c] = z[c]; }
        lam[q * ROWP + 15] = 0.0; zx[q * ROWP + 15] = y;
    }
    double g[16][15], u[16][15], t[16][15], fq[16];
    for (int q = 0; q < 16; q++) lane_columns(q, lam, zx, blk, g[q], u[q], t[q], fq[q]);
    double trash[64];
    // the store order of the kernel: a lane writes ALL 15 entries of its triangular runs; what lies beyond its diagonal
    // lands in a later column's space and is overwritten by the owner, who stores afterwards (descending rows for g; the
    // t runs before the u runs)
    for (int r = 14; r >= 0; r--)
        for (int q = 0; q < 16; q++) out[(q < 15 ? pk(0, q) : pk(0, 30)) + r] = g[q][r];
    for (int r = 0; r < 15; r++)
        for (int q = 0; q < 16; q++) out[(q < 15 ? pk(15, 15 + q) : pk(15, 30)) + r] = t[q][r];
    for (int r = 0; r < 15; r++)
        for (int q = 0; q < 16; q++) (q < 15 ? out + pk(0, 15 + q) : trash)[r] = u[q][r];
    out[pk(30, 30)] = fq[15];
}
extern "C" void hs_hessian(int model, long F, const double *rec, const double *xi, const double *xj, const double *R, double *out) {
    for (long k = 0; k < F; k++) {
        const double *r = rec + k * 87;
        FactorMeas f;
        f.alpha = r; f.beta = r + 3; f.q_KtoK1 = r + 6;
        double lin6[6] = { r[13], r[14], r[15], r[10], r[11], r[12] };
        f.lin = lin6;
        f.J_q = r + 16; f.J_beta = r + 25; f.J_alpha = r + 34; f.H_beta = r + 43; f.H_alpha = r + 52;
        f.dt = r + 61; f.grav = ld3(r + 62); f.q_K_lin = r + 65; f.O_beta = r + 69; f.O_alpha = r + 78;
        f.xi = xi + k * 16; f.xj = xj + k * 16;
        if (model == 1) hessian_factor<1>(f, R + k * 225, out + k * 496); else hessian_factor<2>(f, R + k * 225, out + k * 496);
    }
}
extern "C" void hs_predict(int model, long F, const double *rec, const double *xi, double *xj) {
    for (long k = 0; k < F; k++) {
        const double *r = rec + k * 87, *p = xi + k * 16;
        NavState a; a.q = ldq(p); a.bg = ld3(p + 4); a.v = ld3(p + 7); a.ba = ld3(p + 10); a.p = ld3(p + 13);
        NavState o = (model == 1) ? predict_state<1>(a, ld3(r), ld3(r + 3), ldq(r + 6), r[61], ld3(r + 62))
                                  : predict_state<2>(a, ld3(r), ld3(r + 3), ldq(r + 6), r[61], ld3(r + 62));
        double *d = xj + k * 16;
        d[0] = o.q.x; d[1] = o.q.y; d[2] = o.q.z; d[3] = o.q.w;
        d[4] = o.bg.x; d[5] = o.bg.y; d[6] = o.bg.z; d[7] = o.v.x; d[8] = o.v.y; d[9] = o.v.z;
        d[10] = o.ba.x; d[11] = o.ba.y; d[12] = o.ba.z; d[13] = o.p.x; d[14] = o.p.y; d[15] = o.p.z;
    }
}

extern "C" void hs_forster(long W, int n, const double *kn, const double *lin, const double *sig, double *out) {
    for (long w = 0; w < W; w++) forster_window(n, kn + (size_t)w * (n + 1) * 7, lin + w * 6, sig, out + w * OUTD);
}

// The SO(3) / quaternion helpers of cpi_math.hpp by themselves (same interface as cpi_test_quat_ops / cpi_ref_quat_ops;
// matrices row-major): host emulation of the kernel helpers' LOGIC -- the device instruction sequences (v_rsq_f64 +
// Newton, single-FMA Horner steps) are exercised by tests/test_gpu_quat_ops.py through the real library.
extern "C" int hs_quat_ops(int op, long n, const double *in, double *out) {
    auto put_rm = [](double *o, const M3 &A) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o[i * 3 + j] = A.m[i][j]; };
    auto put_q = [](double *o, Q4 q) { o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w; };
    for (long k = 0; k < n; k++) {
        switch (op) {
            case 0: put_q(out + 4 * k, rot_2_quat(rec_mat(in + 9 * k, 0))); break;
            case 1: put_rm(out + 9 * k, skew(ld3(in + 3 * k))); break;
            case 2: put_rm(out + 9 * k, quat_2_Rot(ldq(in + 4 * k))); break;
            case 3: put_q(out + 4 * k, quat_multiply(ldq(in + 8 * k), ldq(in + 8 * k + 4))); break;
            case 4: put_rm(out + 9 * k, Exp_so3(ld3(in + 3 * k))); break;
            case 5: put_q(out + 4 * k, quat_inv(ldq(in + 4 * k))); break;
            default: return 1;
        }
    }
    return 0;
}
