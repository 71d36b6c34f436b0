// mix_probe.hip (round 6, development tool) -- what does a kernel that does NOTHING but move a sweep's bytes reach?
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/bin/mix_probe tools/exp/mix_probe.hip
//   tools/exp/bin/mix_probe <name> <read bytes per wavefront> <write bytes per wavefront> <wavefronts> [name rd wr n ...]
// A wavefront (one workgroup of 64 lanes) reads `rd` consecutive bytes and writes `wr` consecutive bytes, 16-byte pieces,
// consecutive lanes = consecutive pieces.  Three issue orders:
//   A  loop: load, accumulate; then loop: plain stores                      (k_mixed of store_bw.hip)
//   B  ALL loads of the wavefront in flight before the first store, non-temporal stores (the sweeps' one round trip)
//   C  as B with 4 wavefronts per workgroup (consecutive spans)
// Prints ms per launch and (rd + wr) x wavefronts / time against 8 TB/s.  Used for the copy ceilings of
// profiles/r06_small_sweeps.md: cpi_predict_kernel (64 factors: 13 824 B in / 8 192 B out), the packed sweep (21 factors:
// 16 296 or 19 992 B in / 12 096 B out), the packed-R whitened / Hessian sweeps and cpi_sqrt_info_kernel's packed form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(64) void k_a(const d2 *src, d2 *dst, long long rd16, long long wr16) {
    const long long blk = blockIdx.x;
    double acc = 0;
    for (long long i = threadIdx.x; i < rd16; i += 64) { const d2 v = src[blk * rd16 + i]; acc += v.x + v.y; }
    for (long long i = threadIdx.x; i < wr16; i += 64) dst[blk * wr16 + i] = d2{acc, 2.0 + (double)i};
}
template <int U, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_b(const d2 *src, d2 *dst, long long rd16, long long wr16, long long nw) {
    const long long blk = (long long)blockIdx.x * WPB + (threadIdx.x >> 6);
    if (blk >= nw) return;
    const int lane = threadIdx.x & 63;
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const long long i = lane + 64 * u; v[u] = src[blk * rd16 + (i < rd16 ? i : rd16 - 1)]; }
    double acc = 0;
#pragma unroll
    for (int u = 0; u < U; u++) acc += v[u].x + v[u].y;
    for (long long i = lane; i < wr16; i += 64) __builtin_nontemporal_store(d2{acc, 2.0 + (double)i}, dst + blk * wr16 + i);
}

int main(int argc, char **argv) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int a = 1; a + 4 <= argc; a += 4) {
        const char *name = argv[a];
        const long long rd = atoll(argv[a + 1]), wr = atoll(argv[a + 2]), nw = atoll(argv[a + 3]);
        const long long rd16 = (rd + 15) / 16, wr16 = (wr + 15) / 16;
        d2 *src, *dst;
        if (hipMalloc(&src, (size_t)nw * rd16 * 16) != hipSuccess || hipMalloc(&dst, (size_t)nw * wr16 * 16) != hipSuccess) { printf("%s: out of memory\n", name); return 1; }
        hipMemset(src, 0, (size_t)nw * rd16 * 16);
        const double bytes = (double)nw * (double)(rd + wr);
        auto t = [&](const char *what, auto launch) {
            launch(); hipDeviceSynchronize();
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                for (int r = 0; r < 10; r++) launch();
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms / 10 < best) best = ms / 10;
            }
            printf("%-28s %-34s %9.4f ms  %6.2f TB/s = %.3f of 8 TB/s\n", name, what, best, bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 8e12);
        };
        t("A loop loads, plain stores", [&] { hipLaunchKernelGGL(k_a, dim3((unsigned)nw), dim3(64), 0, 0, src, dst, rd16, wr16); });
        const int U = (int)((rd16 + 63) / 64);
#define CASE(UU) if (U <= UU) { t("B all loads first, nt stores", [&] { hipLaunchKernelGGL((k_b<UU, 1>), dim3((unsigned)nw), dim3(64), 0, 0, src, dst, rd16, wr16, nw); }); \
                                t("C = B, 4 wavefronts / workgroup", [&] { hipLaunchKernelGGL((k_b<UU, 4>), dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, 0, src, dst, rd16, wr16, nw); }); }
        CASE(4) else CASE(8) else CASE(12) else CASE(16) else CASE(24) else printf("%s: read span too long for the unrolled variants\n", name);
#undef CASE
        hipFree(src); hipFree(dst);
    }
    return 0;
}
