// test_facade.cpp -- drives the C++ facade (cpi_host.hpp) exactly like GraphSolver_IMU.cpp:43-75
// drives the reference: read a dumped window, feed_IMU it, print the results for the Python test
// (tests/test_gpu_cpp_facade.py) to compare with the golden vectors.  GPU only.
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../cpi_amd/csrc/cpi_host.hpp"

using namespace cpi_host;

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s <input.bin> <model>\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    double hdr[2];
    if (fread(hdr, 8, 2, f) != 2) return 2;
    const int W = (int)hdr[0], n = (int)hdr[1];
    std::vector<double> kn((size_t)W * (n + 1) * 7), lin((size_t)W * 6), q((size_t)W * 4);
    if (fread(kn.data(), 8, kn.size(), f) != kn.size() || fread(lin.data(), 8, lin.size(), f) != lin.size() ||
        fread(q.data(), 8, q.size(), f) != q.size()) return 2;
    fclose(f);
    const int model = atoi(argv[2]);
    try {
        Context ctx;
        std::vector<CpiBase *> wins;
        CpiBatch batch;
        for (int w = 0; w < W; w++) {
            CpiBase *cpi = (model == 1) ? (CpiBase *)new CpiV1(0.005, 4e-6, 0.01, 2e-4)
                         : (model == 2) ? (CpiBase *)new CpiV2(0.005, 4e-6, 0.01, 2e-4)
                                        : (CpiBase *)new ForsterDiscrete(0.005, 4e-6, 0.01, 2e-4);
            const double *l = &lin[w * 6], *qq = &q[w * 4];
            cpi->setLinearizationPoints({{l[0], l[1], l[2]}}, {{l[3], l[4], l[5]}}, {{qq[0], qq[1], qq[2], qq[3]}}, {{0, 0, 9.8}});
            cpi->imu_avg = false;
            const double *k = &kn[(size_t)w * (n + 1) * 7];
            for (int i = 0; i < n; i++) {
                const double *a = k + 7 * i, *b = k + 7 * (i + 1);
                if (model == 3) {      // GraphSolver_IMU.cpp:171-180: integrateMeasurement(acc, omega, dt)
                    if (b[0] - a[0] >= 0)
                        static_cast<ForsterDiscrete *>(cpi)->integrateMeasurement({{a[4], a[5], a[6]}}, {{a[1], a[2], a[3]}}, b[0] - a[0]);
                } else if (b[0] - a[0] >= 0)  // GraphSolver_IMU.cpp:52
                    cpi->feed_IMU(a[0], b[0], {{a[1], a[2], a[3]}}, {{a[4], a[5], a[6]}}, {{b[1], b[2], b[3]}}, {{b[4], b[5], b[6]}});
            }
            wins.push_back(cpi);
            // three paths: explicit finalize(ctx), NOTHING (window 1: the members compute themselves on first read, on the
            // process's default context -- the caller shape of GraphSolver_IMU.cpp:43-75, no added line), batched
            if (w == 0) cpi->finalize(ctx); else if (w != 1) batch.add(cpi);
        }
        batch.flush(ctx);
        if (W > 1 && model != 3) {
            // window 1 again, written like createimufactor_cpi_v1 / _v2: ctor, setLinearizationPoints, the feed_IMU loop, then the
            // factor straight from the members with the reference's own argument list (GraphSolver_IMU.cpp:74-75 / 129-130) --
            // and a read between two feed_IMU calls sees the state after the calls so far (CpiBase.h:99-124 are live members)
            const double *l = &lin[6], *qq = &q[4], *k = &kn[(size_t)(n + 1) * 7];
            CpiV1 c1(0.005, 4e-6, 0.01, 2e-4);
            CpiV2 c2(0.005, 4e-6, 0.01, 2e-4);
            CpiBase &cpi = (model == 1) ? (CpiBase &)c1 : (CpiBase &)c2;
            cpi.setLinearizationPoints({{l[0], l[1], l[2]}}, {{l[3], l[4], l[5]}}, {{qq[0], qq[1], qq[2], qq[3]}}, {{0, 0, 9.8}});
            cpi.imu_avg = false;
            double dt_half = -1, dt_sum = 0;
            for (int i = 0; i < n; i++) {
                const double *a = k + 7 * i, *b = k + 7 * (i + 1);
                if (b[0] - a[0] >= 0) { cpi.feed_IMU(a[0], b[0], {{a[1], a[2], a[3]}}, {{a[4], a[5], a[6]}}, {{b[1], b[2], b[3]}}, {{b[4], b[5], b[6]}}); dt_sum += b[0] - a[0]; }
                if (i == n / 2) { dt_half = cpi.DT; if (!(dt_half > 0) || dt_half > dt_sum + 1e-12 || dt_half < dt_sum - 1e-12) { fprintf(stderr, "mid-window read: DT %.17g, fed %.17g\n", dt_half, dt_sum); return 1; } }
            }
            ImuFactorCPI fac = (model == 1)
                ? ImuFactorCPI(cpi.P_meas, cpi.DT, cpi.grav, cpi.alpha_tau, cpi.beta_tau, cpi.q_k2tau, cpi.b_a_lin, cpi.b_w_lin, cpi.J_q, cpi.J_b,
                               cpi.J_a, cpi.H_b, cpi.H_a)
                : ImuFactorCPI(cpi.P_meas, cpi.DT, cpi.grav, cpi.alpha_tau, cpi.beta_tau, cpi.q_k2tau, cpi.q_k_lin, cpi.b_a_lin, cpi.b_w_lin, cpi.J_q,
                               cpi.J_b, cpi.J_a, cpi.H_b, cpi.H_a, cpi.O_b, cpi.O_a);
            ImuFactorCPI ref(*wins[1]);
            double xi[16] = { qq[0], qq[1], qq[2], qq[3], l[0], l[1], l[2], 0.3, -0.2, 0.1, l[3], l[4], l[5], 1, 2, 3 };
            double xj[16] = { qq[0], qq[1], qq[2], qq[3], l[0] + 1e-3, l[1], l[2] - 2e-3, 0.35, -0.1, 0.1, l[3], l[4] + 1e-2, l[5], 1.1, 2.2, 2.9 };
            double e1[15], e2[15], Ha[225], Hb[225], Hc[225], Hd[225];
            fac.evaluateError(ctx, xi, xj, e1, Ha, Hb);
            ref.evaluateError(ctx, xi, xj, e2, Hc, Hd);
            for (int i = 0; i < 15; i++) if (e1[i] != e2[i]) { fprintf(stderr, "reference-shaped factor: err[%d] %.17g vs %.17g\n", i, e1[i], e2[i]); return 1; }
            for (int i = 0; i < 225; i++) if (Ha[i] != Hc[i] || Hb[i] != Hd[i]) { fprintf(stderr, "reference-shaped factor: H[%d] differs\n", i); return 1; }
            CpiBase copy(cpi);                                       // copies re-bind the lazy members to the copy
            copy.feed_IMU(k[7 * n], k[7 * n] + 0.005, {{0, 0, 0}}, {{0, 0, 9.8}});
            if (!((double)copy.DT > (double)cpi.DT)) { fprintf(stderr, "copy: DT %.17g vs %.17g\n", (double)copy.DT, (double)cpi.DT); return 1; }
            printf("SHAPE ok %.17g\n", dt_half);
        }
        for (int w = 0; w < W; w++) {
            const CpiBase &c = *wins[w];
            printf("%.17g", (double)c.DT);
            for (double v : c.alpha_tau) printf(" %.17g", v);
            for (double v : c.beta_tau) printf(" %.17g", v);
            for (double v : c.q_k2tau) printf(" %.17g", v);
            for (double v : c.J_q) printf(" %.17g", v);
            for (double v : c.J_a) printf(" %.17g", v);
            for (double v : c.J_b) printf(" %.17g", v);
            for (double v : c.H_a) printf(" %.17g", v);
            for (double v : c.H_b) printf(" %.17g", v);
            for (double v : c.O_a) printf(" %.17g", v);
            for (double v : c.O_b) printf(" %.17g", v);
            for (double v : c.P_meas) printf(" %.17g", v);
            printf("\n");
        }
        // one evaluateError at the predicted-like state pair (state_i == state_j): bias rows must vanish
        ImuFactorCPI fac(*wins[0]);
        double xi[16] = { q[0], q[1], q[2], q[3], lin[0], lin[1], lin[2], 0.3, -0.2, 0.1, lin[3], lin[4], lin[5], 1, 2, 3 };
        double err[15], H1[225], H2[225];
        fac.evaluateError(ctx, xi, xi, err, H1, H2);
        printf("ERR");
        for (double v : err) printf(" %.17g", v);
        printf("\n");
        // mean-only flush (models 1 / 2): the recorded windows go straight into the tiled layout with their own interval
        // counts (skipped dt < 0 intervals make the golden windows differ in length)
        if (model != 3) {
            CpiBatch means;
            for (int w = 0; w < W; w++) { wins[w]->DT = -1; means.add(wins[w]); }
            means.flush_means(ctx);
            for (int w = 0; w < W; w++) {
                const CpiBase &c = *wins[w];
                printf("MEAN %.17g", (double)c.DT);
                for (double v : c.alpha_tau) printf(" %.17g", v);
                for (double v : c.beta_tau) printf(" %.17g", v);
                for (double v : c.q_k2tau) printf(" %.17g", v);
                printf("\n");
            }
            // after a mean-only flush the Jacobian / covariance members are NOT valid for the recorded intervals -- a
            // read of one runs the whole window (the means read above did not); poisoned here so that a stale read would show.
            // And two threads reading DIFFERENT preintegrators lazily use their own (thread-local) default contexts.
            CpiBase &c0 = *wins[0], &c1 = *wins[W > 2 ? 2 : 0];
            CpiResult poisoned = c0.result();                      // full results (finalize)
            {
                CpiBatch again;
                again.add(&c0); again.add(&c1);
                again.flush_means(ctx);                            // means only: the full members are stale from here on
            }
            double tr[2] = {0, 0}, jq0[2] = {0, 0};
            std::thread t0([&] { for (int i = 0; i < 15; i++) tr[0] += c0.P_meas[i * 16]; jq0[0] = c0.J_q[0]; });
            std::thread t1([&] { for (int i = 0; i < 15; i++) tr[1] += c1.P_meas[i * 16]; jq0[1] = c1.J_q[0]; });
            t0.join(); t1.join();
            double want = 0;
            for (int i = 0; i < 15; i++) want += poisoned.P_meas[i * 16];
            printf("AFTERMEANS %s %.17g %.17g %.17g %.17g\n", (tr[0] == want && jq0[0] == poisoned.J_q[0] && tr[1] > 0) ? "ok" : "STALE", tr[0], want, tr[1], jq0[1]);
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "cpi_host error: %s\n", e.what());
        return 1;
    }
    return 0;
}
