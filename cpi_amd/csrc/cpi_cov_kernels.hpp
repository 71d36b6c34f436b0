// cpi_cov_kernels.hpp -- covariance (+ state transition) recursion of CPI models 1 / 2 and the Forster comparator.
// Part of the translation unit cpi_cov.hip (included there after cpi_math.hpp / cpi_device_util.hpp; not a stand-alone header).
#pragma once

namespace {

// ============================================================================================
// covariance (+ state transition) kernel
// ============================================================================================
// Two co-resident wavefronts per SIMD hide the LDS exchange latency of the recursion: <= 256 registers and
// <= 20 KB of LDS per wavefront.  The latter is why a phase-A pass stages GROUP/2 intervals per window
// (half the lanes take part in it); measured on MI355X against the one-wave-per-SIMD variant:
// model 2 5.0 -> 3.8 ms, model 1 2.27 -> 2.0 ms per 100 k windows.
#ifndef CPI_COV_WPS
#define CPI_COV_WPS 2
#endif
#ifndef CPI_FORSTER_MEAN_LANES
#define CPI_FORSTER_MEAN_LANES 6
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
    // windows cut out of a stream in flight (PreArgs::tstart / tend): see the knot loads of phase A
    const bool cut = A.tstart != nullptr;
    const bool tail = cut && (A.tend[w] == A.tend[w]) && (A.count[w] <= A.N);

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
                if (!cut) {
#pragma unroll
                    for (int i = 0; i < 14; i++) a[i] = ka[i];
                } else {
                    // the closing knot of a tail interval does not exist in memory: the opening reading held until tend
                    const bool tl = tail && s == n - 1;
                    const double *kb = tl ? ka : ka + 7;
#pragma unroll
                    for (int i = 0; i < 7; i++) { a[i] = ka[i]; a[7 + i] = kb[i]; }
                    if (s == 0) a[0] = A.tstart[wq];
                    if (tl) a[7] = A.tend[wq];
                }
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
                // From here to the end of the stage the wavefront sits on the latency chain "exchange write -> transposed
                // read -> accumulate": it issues ahead of its SIMD neighbour, which is in the F x phase of another stage
                // and has independent arithmetic to fill the gaps with.  Measured (same box, alternating runs, 100 k x 50):
                // V1 full 1.357 -> 1.321 ms (-2.7 %), V2 full 2.719 -> 2.701 ms (-0.7 %); the opposite assignment (priority
                // during F x) loses 1.5 / 3 %, a constant priority changes nothing.
                __builtin_amdgcn_s_setprio(3);
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
                    for (int i = 0; i < D::NR; i++)
                        mt[i] = (MODEL == 1) ? dpp_shr6_bank3(ex_row[exch_pos<MODEL>(i)], Xs[i])
                                             : dpp_shr6_bank3_oddrows(ex_row[exch_pos<MODEL>(i)], Xs[i]);
                    cov_stage_finish_regs(Ln, stg, M, mt);
                } else {
                    cov_stage_finish(Ln, stg, M, ex_row);
                }
                __builtin_amdgcn_s_setprio(0);
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
    if (A.out.P_sym && jj < 15) {   // the same column, rows 0 .. jj only: the packed upper triangle (CPI_TRI_INDEX), 960 B per window
        double *p = A.out.P_sym + w * CPI_TRI_DOUBLES + jj * (jj + 1) / 2;
#pragma unroll
        for (int i = 0; i < 15; i++) if (i <= jj) p[i] = Ln.P0[i];
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
// Jacobians, lanes 3-5 column j-3 of the two accelerometer-bias Jacobians; those six lanes carry the means (until late in
// round 6 every lane did -- "the SIMD cost is the same as one lane doing it": the instruction count is, the power is not, and
// this is the most FP64-dense kernel of the library: 0.777 -> 0.725 ms per 100 k x 50 with ten lanes of sixteen sitting the mean
// and Jacobian steps out; CPI_FORSTER_MEAN_LANES).  P' = F P F^T + G per interval: F x is lane-local (F is sparse), the
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
    // windows cut out of a stream in flight (PreArgs::tstart / tend): the stamps of the first knot and of a tail interval's
    // closing knot (which does not exist in memory) are patched where the knots are loaded
    const bool cut = A.tstart != nullptr;
    const bool tail = cut && (A.tend[w] == A.tend[w]) && (A.count[w] <= A.N);
    auto load_knot = [&](double (&k)[8], int s) {      // reading of knot s and the stamp of knot s + 1
        const double *ka = A.knots + (k0 + s) * 7;
#pragma unroll
        for (int i = 0; i < 7; i++) k[i] = ka[i];
        if (!cut) { k[7] = ka[7]; return; }
        const bool tl = tail && s == n - 1;
        const double tnext = ka[tl ? 0 : 7];        // a tail interval has no closing knot in memory: the address stays inside
        k[7] = tl ? A.tend[w] : tnext;
        if (s == 0) k[0] = A.tstart[w];
    };

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
    if (j < CH && j < n) load_knot(kn, j);
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
            if (s + CH < n) load_knot(kn, s + CH);
        }
        wave_lds_fence();

        // ---- phase C: the sequential recursion over the staged intervals
        const int cnt = min(CH, nmax - base);
        for (int sl = 0; sl < cnt; ++sl) {
            const double *ir = irs + (g * CH + sl) * IRD;   // group-uniform address: LDS broadcast
            const fsd::Rec r = fsd::get_rec(ir);
            if (j < CPI_FORSTER_MEAN_LANES) {               // lanes 0-5 carry the Jacobian columns (and lane 0 stores the means): the
                fsd::jac_step(J, m.R, r, ek, eg);           // others sit these ~80 FP64 instructions out -- same instruction stream for the
                fsd::mean_step(m, r);                       // wavefront, 6 / 16 of the lanes switching (profiles/r06_small_sweeps.md section 6)
            }                                               // (jac_step uses the rotation BEFORE this interval)
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
    if (A.out.P_sym && j < 15) {   // packed upper triangle of the same matrix
        double *p = A.out.P_sym + w * CPI_TRI_DOUBLES + j * (j + 1) / 2;
#pragma unroll
        for (int i = 0; i < 15; i++) if (i <= j) p[i] = x[i];
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


}  // namespace
