// test_facade_lazy_cpu.cpp -- the C++ facade's lazy result members WITHOUT a GPU (tests/test_cpp_facade_cpu.py):
// fresh preintegrators read as the reference's freshly constructed ones (CpiBase.h:99-124: zeros, q = (0,0,0,1)) without touching
// the device; stored results are read back unchanged; copies own their members; and a read that NEEDS the device -- intervals were
// recorded, nothing has run -- fails loudly on a machine without one (no CPU fallback), instead of returning stale values.
#include <cstdio>
#include <stdexcept>

#include "../../cpi_amd/csrc/cpi_host.hpp"

using namespace cpi_host;

static double first(const Vec3 &v) { return v[0]; }                 // reference-shaped consumers: by const reference / by value
static double trace(const Mat15 &P) { double t = 0; for (int i = 0; i < 15; i++) t += P[i * 16]; return t; }

int main() {
    CpiV1 a(0.005, 4e-6, 0.01, 2e-4);
    a.setLinearizationPoints({{1e-3, 0, 0}}, {{0, 2e-2, 0}}, {{0, 0, 0, 1}}, {{0, 0, 9.8}});
    // nothing recorded: no device is needed, the members are what the reference's constructor leaves
    const double dt0 = a.DT;
    if (dt0 != 0.0 || first(a.alpha_tau) != 0.0 || a.q_k2tau[3] != 1.0 || trace(a.P_meas) != 0.0 || a.J_q.size() != 9) { printf("FAIL fresh\n"); return 1; }
    // stored results (what CpiBatch::flush does) are read back without a device
    CpiResult r;
    r.DT = 0.25; r.alpha_tau = Vec3{{1, 2, 3}}; r.P_meas[16] = 4.0;
    a.set_result(r);
    if ((double)a.DT != 0.25 || first(a.alpha_tau) != 1.0 || trace(a.P_meas) != 4.0) { printf("FAIL stored\n"); return 1; }
    double sum = 0;
    for (double v : a.alpha_tau) sum += v;
    if (sum != 6.0 || a.alpha_tau.data()[2] != 3.0) { printf("FAIL iterate\n"); return 1; }
    // copies own their members (the lazy fields are bound to the copy, not to the original)
    CpiV1 b(a);
    b.DT = 0.5;
    if ((double)a.DT != 0.25 || (double)b.DT != 0.5 || first(b.alpha_tau) != 1.0) { printf("FAIL copy\n"); return 1; }
    CpiV1 c(0.1, 0.1, 0.1, 0.1);
    c = a;
    if ((double)c.DT != 0.25 || c.b_a_lin[1] != 2e-2) { printf("FAIL assign\n"); return 1; }
    // a pending window: the read must go to the device -- and on a machine without one it throws
    a.feed_IMU(0.0, 0.005, {{0.1, 0, 0}}, {{0, 0, 9.8}});
    try {
        const double dt = a.DT;
        printf("DEVICE dt=%.17g\n", dt);                             // a GPU is present: the window ran
        return (dt == 0.005) ? 0 : 1;
    } catch (const std::runtime_error &e) {
        printf("THROWS %s\n", e.what());
        return 0;
    }
}
