// test_imu_stream.cpp -- GPU: a caller shaped like GraphSolver::createimufactor_cpi_v1 / _v2 (GraphSolver_IMU.cpp:34-134)
// for every update time at once: the IMU text file goes through parse_imu_text into an ImuStream, ImuStream::preintegrate
// cuts and preintegrates all windows on the device (cpi_preintegrate_stream_host), and the first window's result is wrapped
// in an ImuFactorCPI and evaluated.  Prints one line of 299 numbers per window (DT alpha beta q J_q J_a J_b H_a H_b O_a O_b
// P), a COUNT line and an ERR line for the Python test to compare with the oracle's restatement of the reference's deque loop.
//   test_imu_stream <imu text file> <update times file> <lin file: U x 10 doubles b_w b_a q_k_lin> <model>
#include <cstdio>
#include <fstream>
#include <sstream>

#include "../../cpi_amd/csrc/cpi_host.hpp"

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    using namespace cpi_host;
    std::ifstream f(argv[1]);
    std::stringstream ss; ss << f.rdbuf();
    ImuStream imu;
    {
        const std::vector<double> k = parse_imu_text(ss.str());
        for (size_t i = 0; i + 6 < k.size(); i += 7) imu.push(k[i], Vec3{{k[i + 1], k[i + 2], k[i + 3]}}, Vec3{{k[i + 4], k[i + 5], k[i + 6]}});
    }
    std::vector<double> ut, lin, qk;
    { std::ifstream g(argv[2]); double t; while (g >> t) ut.push_back(t); }
    {
        std::ifstream g(argv[3]);
        double v[10];
        while (g >> v[0] >> v[1] >> v[2] >> v[3] >> v[4] >> v[5] >> v[6] >> v[7] >> v[8] >> v[9]) {
            lin.insert(lin.end(), v, v + 6);
            qk.insert(qk.end(), v + 6, v + 10);
        }
    }
    const int model = atoi(argv[4]);
    try {
        Context ctx;
        CpiV1 proto1(0.005, 4e-6, 0.01, 2e-4);      // cpi_compare/launch/synthetic_test.launch:13-17
        CpiV2 proto2(0.005, 4e-6, 0.01, 2e-4);
        ForsterDiscrete proto3(0.005, 4e-6, 0.01, 2e-4);
        CpiBase &proto = model == 2 ? (CpiBase &)proto2 : (model == 3 ? (CpiBase &)proto3 : (CpiBase &)proto1);
        proto.grav = Vec3{{0, 0, 9.8}};
        cpi_params prm = proto.params();
        std::vector<int32_t> counts;
        std::vector<CpiResult> res = imu.preintegrate(ctx, prm, ut, lin, qk, &counts);
        for (size_t w = 0; w < res.size(); w++) {
            const CpiResult &r = res[w];
            printf("%.17g", r.DT);
            for (double v : r.alpha_tau) printf(" %.17g", v);
            for (double v : r.beta_tau) printf(" %.17g", v);
            for (double v : r.q_k2tau) printf(" %.17g", v);
            const Mat3 *ms[7] = { &r.J_q, &r.J_a, &r.J_b, &r.H_a, &r.H_b, &r.O_a, &r.O_b };
            for (const Mat3 *m : ms) for (double v : *m) printf(" %.17g", v);
            for (double v : r.P_meas) printf(" %.17g", v);
            printf("\n");
        }
        printf("COUNT");
        for (int32_t c : counts) printf(" %d", c);
        printf("\n");
        // a window too long for the bound the caller gave is an error, not a truncated result
        bool threw = false;
        try { imu.preintegrate(ctx, prm, ut, lin, qk, nullptr, 3); } catch (const std::runtime_error &) { threw = true; }
        printf("BOUND %d\n", threw ? 1 : 0);
        // the factor of window 1 between two states (identity attitude; biases at the linearisation point)
        if (res.size() > 1 && model != 3) {
            ImuFactorCPI fac(model, res[1], proto.grav, Vec3{{lin[6], lin[7], lin[8]}}, Vec3{{lin[9], lin[10], lin[11]}}, Vec4{{qk[4], qk[5], qk[6], qk[7]}});
            double xi[16] = { 0, 0, 0, 1, lin[6], lin[7], lin[8], 0.1, -0.2, 0.05, lin[9], lin[10], lin[11], 1, 2, 3 };
            double xj[16] = { 0, 0, 0, 1, lin[6], lin[7], lin[8], 0.1, -0.2, 0.05, lin[9], lin[10], lin[11], 1, 2, 3 };
            double e[15], H1[225], H2[225];
            fac.evaluateError(ctx, xi, xj, e, H1, H2);
            printf("ERR");
            for (double v : e) printf(" %.17g", v);
            printf("\n");
        }
    } catch (const std::exception &ex) {
        fprintf(stderr, "test_imu_stream: %s\n", ex.what());
        return 1;
    }
    return 0;
}
