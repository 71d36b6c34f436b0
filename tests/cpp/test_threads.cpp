// Thread-safety of the C-ABI as INTEGRATION.md states it: a context is owned by one thread at a time, different contexts
// are independent.  T host threads, one context each, run the same batch concurrently -- host-pointer entry (pipelined
// staging owned by the context), then the factor sweep on the device -- and every thread must reproduce the result of a
// single-threaded run BIT FOR BIT.  usage: test_threads <input.bin> <threads>; input as for test_group.  Compiled with g++.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/cpi_amd.h"

struct Result { std::vector<double> DT, alpha, beta, q, P, err; int rc = 0; };

static int run_once(int64_t W, int32_t N, const std::vector<double> &kn, const std::vector<double> &lin, const std::vector<double> &qk,
                    int reps, Result &r) {
    cpi_ctx *ctx = nullptr;
    if (cpi_ctx_create(0, nullptr, &ctx) != CPI_OK) return 10;
    cpi_params prm{};
    prm.sigma_w = 0.005; prm.sigma_wb = 4e-6; prm.sigma_a = 0.01; prm.sigma_ab = 2e-4;
    prm.grav[2] = 9.8; prm.model = CPI_MODEL_V2; prm.state_transition_jacobians = 1; prm.lanes_per_window = 1;
    r.DT.assign(W, 0); r.alpha.assign(W * 3, 0); r.beta.assign(W * 3, 0); r.q.assign(W * 4, 0); r.P.assign(W * 225, 0);
    std::vector<double> J[7];
    for (auto &j : J) j.assign(W * 9, 0);
    cpi_outputs o{};
    o.DT = r.DT.data(); o.alpha = r.alpha.data(); o.beta = r.beta.data(); o.q = r.q.data(); o.P = r.P.data();
    o.J_q = J[0].data(); o.J_a = J[1].data(); o.J_b = J[2].data(); o.H_a = J[3].data(); o.H_b = J[4].data(); o.O_a = J[5].data(); o.O_b = J[6].data();
    int rc = 0;
    for (int i = 0; i < reps && rc == 0; i++)
        rc = cpi_preintegrate_batch_host(ctx, &prm, W, N, kn.data(), nullptr, nullptr, 0, lin.data(), qk.data(), &o);
    if (rc != 0) { std::fprintf(stderr, "preintegrate: %s\n", cpi_last_error(ctx)); cpi_ctx_destroy(ctx); return 11; }
    // the factor sweep of the same windows against a chain of W + 1 states (host-pointer entry)
    std::vector<double> st((size_t)(W + 1) * 16, 0.0);
    for (int64_t s = 0; s <= W; s++) { st[s * 16 + 3] = 1.0; st[s * 16 + 7] = 0.01 * (double)(s % 7); st[s * 16 + 13] = 0.1 * (double)(s % 5); }
    r.err.assign(W * 15, 0);
    const double grav[3] = { 0, 0, 9.8 };
    for (int i = 0; i < reps && rc == 0; i++)
        rc = cpi_factor_eval_batch_host(ctx, CPI_MODEL_V2, grav, W, &o, lin.data(), qk.data(), st.data(), W + 1, nullptr, nullptr, r.err.data(), nullptr, nullptr);
    if (rc != 0) std::fprintf(stderr, "factor: %s\n", cpi_last_error(ctx));
    cpi_ctx_destroy(ctx);
    return rc ? 12 : 0;
}

int main(int argc, char **argv) {
    if (argc < 3) return 1;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 1;
    double hdr[2];
    if (std::fread(hdr, 8, 2, f) != 2) return 1;
    const int64_t W = (int64_t)hdr[0];
    const int32_t N = (int32_t)hdr[1];
    std::vector<double> kn((size_t)W * (N + 1) * 7), lin((size_t)W * 6), qk((size_t)W * 4);
    if (std::fread(kn.data(), 8, kn.size(), f) != kn.size() || std::fread(lin.data(), 8, lin.size(), f) != lin.size() ||
        std::fread(qk.data(), 8, qk.size(), f) != qk.size()) return 1;
    std::fclose(f);
    const int T = std::atoi(argv[2]);
    Result ref;
    if (int rc = run_once(W, N, kn, lin, qk, 1, ref)) return rc;
    std::vector<Result> res(T);
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&, t] { res[t].rc = run_once(W, N, kn, lin, qk, 3, res[t]); });
    for (auto &x : th) x.join();
    for (int t = 0; t < T; t++) {
        if (res[t].rc) { std::fprintf(stderr, "thread %d failed: %d\n", t, res[t].rc); return 20; }
        auto same = [](const std::vector<double> &a, const std::vector<double> &b) { return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * 8) == 0; };
        if (!same(res[t].DT, ref.DT) || !same(res[t].alpha, ref.alpha) || !same(res[t].beta, ref.beta) || !same(res[t].q, ref.q) ||
            !same(res[t].P, ref.P) || !same(res[t].err, ref.err)) { std::fprintf(stderr, "thread %d differs from the single-threaded run\n", t); return 21; }
    }
    double s = 0;
    for (double v : ref.err) s += v * v;
    std::printf("threads ok T=%d W=%lld |err|^2=%.6e\n", T, (long long)W, s);
    return 0;
}
