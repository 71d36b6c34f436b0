// Single-process multi-GPU host written against the C-ABI device-set entries (include/cpi_amd.h, cpi_host::DeviceGroup):
// the shape INTEGRATION.md shows for a C++ caller that shards a batch of windows over the GPUs of a node and gathers
// the outputs on GPU 0.  usage: test_group <input.bin> <model> [ngpus]   -- input: {W, N} then knots[W][N+1][7], lin[W][6],
// q_k_lin[W][4] (doubles).  Prints one line per window: DT alpha(3) beta(3) q(4) P(225), gathered on the root.
// A 5th argument k > 0 runs the exchange INSIDE the batch (DeviceGroup::gather_chunk, ABI 3): every rank's block in k sub-blocks, each
// computed into a slab of its own that carries the covariance as its packed upper triangle (cpi_outputs.P_sym); the host unpacks it
// (CPI_TRI_INDEX) before printing, so the text equals the dense run's.
// Compiled with g++ (no hipcc): only the HIP runtime API for memory is needed.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include <cstring>

#include "../../cpi_amd/csrc/cpi_host.hpp"
#ifdef CPI_TEST_HOOKS   // the "shared" mode (n ranks on ONE device) needs libcpi_amd_test.so; without it this host links the product library
#include "../../include/cpi_amd_test.h"
#endif

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

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
    const int model = std::atoi(argv[2]);
    int ndev = 0;
    HIP_OK(hipGetDeviceCount(&ndev));
    const int n = argc > 3 ? std::atoi(argv[3]) : ndev;

    const bool shared = argc > 4 && std::strcmp(argv[4], "shared") == 0;
    cpi_group *raw = nullptr;
#ifdef CPI_TEST_HOOKS
    if (shared && cpi_test_group_create_shared(n, 0, &raw) != CPI_OK) { std::fprintf(stderr, "%s\n", cpi_group_last_error(nullptr)); return 4; }
#else
    if (shared) { std::fprintf(stderr, "the shared-device mode is a test hook: compile with -DCPI_TEST_HOOKS against libcpi_amd_test.so\n"); return 4; }
#endif
    cpi_host::DeviceGroup grp = shared ? cpi_host::DeviceGroup(raw) : cpi_host::DeviceGroup(n);
    cpi_params prm{};
    prm.sigma_w = 0.005; prm.sigma_wb = 4e-6; prm.sigma_a = 0.01; prm.sigma_ab = 2e-4;
    prm.grav[2] = 9.8; prm.model = model; prm.state_transition_jacobians = 1;

    const int chunks = argc > 5 ? std::atoi(argv[5]) : 0;
    if (chunks > 0) {
        HIP_OK(hipSetDevice(0));
        cpi_outputs root{};
        HIP_OK(hipMalloc((void **)&root.DT, (size_t)W * 8)); HIP_OK(hipMalloc((void **)&root.alpha, (size_t)W * 24));
        HIP_OK(hipMalloc((void **)&root.beta, (size_t)W * 24)); HIP_OK(hipMalloc((void **)&root.q, (size_t)W * 32));
        HIP_OK(hipMalloc((void **)&root.P_sym, (size_t)W * CPI_TRI_DOUBLES * 8));
        double dummy = 0;
        cpi_outputs mask{};
        mask.DT = mask.alpha = mask.beta = mask.q = mask.P_sym = &dummy;
        int messages = 0;
        for (int c = 0; c < chunks; c++) {
            std::vector<cpi_outputs> loc(n);
            int64_t cper = 0;
            { int64_t a, b; grp.chunk_bounds(W, 0, 0, chunks, a, b); cper = b - a; }      // the common sub-block size (rank 0, chunk 0 is never short unless W is tiny)
            for (int r = 0; r < n; r++) {
                int64_t lo, hi;
                grp.chunk_bounds(W, r, c, chunks, lo, hi);
                const int64_t w = hi - lo;
                if (w <= 0) continue;
                HIP_OK(hipSetDevice(shared ? 0 : r));
                double *dk, *dl, *dq, *slab;
                HIP_OK(hipMalloc((void **)&dk, (size_t)w * (N + 1) * 56)); HIP_OK(hipMalloc((void **)&dl, (size_t)w * 48)); HIP_OK(hipMalloc((void **)&dq, (size_t)w * 32));
                HIP_OK(hipMemcpy(dk, kn.data() + (size_t)lo * (N + 1) * 7, (size_t)w * (N + 1) * 56, hipMemcpyHostToDevice));
                HIP_OK(hipMemcpy(dl, lin.data() + (size_t)lo * 6, (size_t)w * 48, hipMemcpyHostToDevice));
                HIP_OK(hipMemcpy(dq, qk.data() + (size_t)lo * 4, (size_t)w * 32, hipMemcpyHostToDevice));
                const int64_t wb = cper > w ? cper : w;
                HIP_OK(hipMalloc((void **)&slab, cpi_outputs_slab_doubles(&mask, wb) * 8));
                if (cpi_outputs_bind_slab(&mask, wb, slab, &loc[r]) != CPI_OK) return 5;
                if (cpi_preintegrate_batch(grp.ctx(r), &prm, w, N, dk, nullptr, nullptr, dl, dq, &loc[r]) != CPI_OK) {
                    std::fprintf(stderr, "rank %d: %s\n", r, cpi_last_error(grp.ctx(r)));
                    return 3;
                }
            }
            grp.gather_chunk(0, W, c, chunks, loc.data(), root);      // sub-block c on the exchange streams; the kernels of c + 1 follow at once
            messages += grp.last_gather_messages();
        }
        grp.synchronize();
        HIP_OK(hipSetDevice(0));
        std::vector<double> DT(W), al(W * 3), be(W * 3), q(W * 4), Ps((size_t)W * CPI_TRI_DOUBLES);
        HIP_OK(hipMemcpy(DT.data(), root.DT, W * 8, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(al.data(), root.alpha, W * 24, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(be.data(), root.beta, W * 24, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(q.data(), root.q, W * 32, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(Ps.data(), root.P_sym, (size_t)W * CPI_TRI_DOUBLES * 8, hipMemcpyDeviceToHost));
        for (int64_t w = 0; w < W; w++) {
            std::printf("%.17g", DT[w]);
            for (int i = 0; i < 3; i++) std::printf(" %.17g", al[w * 3 + i]);
            for (int i = 0; i < 3; i++) std::printf(" %.17g", be[w * 3 + i]);
            for (int i = 0; i < 4; i++) std::printf(" %.17g", q[w * 4 + i]);
            for (int j = 0; j < 15; j++)                                  // dense column-major from the packed upper triangle
                for (int i = 0; i < 15; i++) std::printf(" %.17g", Ps[(size_t)w * CPI_TRI_DOUBLES + (i <= j ? CPI_TRI_INDEX(i, j) : CPI_TRI_INDEX(j, i))]);
            std::printf("\n");
        }
        std::fprintf(stderr, "group of %d %s, %lld windows gathered on rank 0 in %d sub-blocks, %d message(s) per peer in all\n", n,
                     shared ? "rank(s) on one device" : "device(s)", (long long)W, chunks, messages);
        return 0;
    }

    static const int FN[5] = { 1, 3, 3, 4, 225 };                 // DT alpha beta q P
    std::vector<cpi_outputs> local(n);
    std::vector<std::vector<void *>> owned(n);
    cpi_outputs root{};
    const int root_rank = 0;
    for (int r = 0; r < n; r++) {
        int64_t lo, hi;
        grp.bounds(W, r, lo, hi);
        const int64_t w = hi - lo;
        int dev = -1;
        // the group's contexts own devices 0 .. n-1 here (devices == nullptr); "shared": all on device 0
        dev = shared ? 0 : r;
        HIP_OK(hipSetDevice(dev));
        double *dk = nullptr, *dl = nullptr, *dq = nullptr, *out[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
        if (w > 0) {
            HIP_OK(hipMalloc((void **)&dk, (size_t)w * (N + 1) * 56)); HIP_OK(hipMalloc((void **)&dl, (size_t)w * 48)); HIP_OK(hipMalloc((void **)&dq, (size_t)w * 32));
            HIP_OK(hipMemcpy(dk, kn.data() + (size_t)lo * (N + 1) * 7, (size_t)w * (N + 1) * 56, hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(dl, lin.data() + (size_t)lo * 6, (size_t)w * 48, hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(dq, qk.data() + (size_t)lo * 4, (size_t)w * 32, hipMemcpyHostToDevice));
        }
        cpi_outputs o{};
        if (w > 0) {   // one slab for the five wanted fields (mask: any non-NULL pointer marks a field as wanted)
            double dummy = 0;
            cpi_outputs mask{};
            mask.DT = mask.alpha = mask.beta = mask.q = mask.P = &dummy;
            HIP_OK(hipMalloc((void **)&out[0], cpi_outputs_slab_doubles(&mask, w) * 8));
            if (cpi_outputs_bind_slab(&mask, w, out[0], &o) != CPI_OK) return 5;
        }
        local[r] = o;
        owned[r] = { dk, dl, dq, out[0], out[1], out[2], out[3], out[4] };
        if (r == root_rank) {
            double *ro[5];
            for (int k = 0; k < 5; k++) HIP_OK(hipMalloc((void **)&ro[k], (size_t)W * FN[k] * 8));
            root.DT = ro[0]; root.alpha = ro[1]; root.beta = ro[2]; root.q = ro[3]; root.P = ro[4];
        }
        // rank r's block on its own GPU, asynchronously on the group's stream of that GPU
        if (w > 0 && cpi_preintegrate_batch(grp.ctx(r), &prm, w, N, dk, nullptr, nullptr, dl, dq, &o) != CPI_OK) {
            std::fprintf(stderr, "rank %d: %s\n", r, cpi_last_error(grp.ctx(r)));
            return 3;
        }
    }
    grp.gather(root_rank, W, local.data(), root);                 // the one exchange step: slabs -> GPU 0
    grp.synchronize();
    HIP_OK(hipSetDevice(0));
    std::vector<double> DT(W), al(W * 3), be(W * 3), q(W * 4), P((size_t)W * 225);
    HIP_OK(hipMemcpy(DT.data(), root.DT, W * 8, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(al.data(), root.alpha, W * 24, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(be.data(), root.beta, W * 24, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(q.data(), root.q, W * 32, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(P.data(), root.P, (size_t)W * 1800, hipMemcpyDeviceToHost));
    for (int64_t w = 0; w < W; w++) {
        std::printf("%.17g", DT[w]);
        for (int i = 0; i < 3; i++) std::printf(" %.17g", al[w * 3 + i]);
        for (int i = 0; i < 3; i++) std::printf(" %.17g", be[w * 3 + i]);
        for (int i = 0; i < 4; i++) std::printf(" %.17g", q[w * 4 + i]);
        for (int i = 0; i < 225; i++) std::printf(" %.17g", P[(size_t)w * 225 + i]);
        std::printf("\n");
    }
    std::fprintf(stderr, "group of %d %s, %lld windows gathered on rank %d, %d message(s) per peer\n", n, shared ? "rank(s) on one device" : "device(s)",
                 (long long)W, root_rank, grp.last_gather_messages());
    return 0;
}
