// Where is a re-fetched line served from, and at what rate?  (development tool, round 5: profiles/r05_mean_traffic.md)
// No rocprofv3 counter on gfx950 separates Infinity-Cache hits from DRAM reads (both leave the L2 through TCC_EA0_RDREQ), so the
// rate is measured instead: a linear 16-B-per-lane read of a buffer of S bytes, repeated -- S <= 32 MiB is served by the L2s,
// S <= 256 MiB by the Infinity Cache, larger S by HBM -- and the same bytes read as TWO passes a distance D apart (every line is
// requested twice, D bytes of other traffic in between: the mean kernel's half-read lines come back ~20 MB of traffic later).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/bin/mall_probe tools/exp/mall_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));

// a wavefront owns `span16` consecutive 16-byte pieces (4 loads in flight per lane and trip)
__global__ __launch_bounds__(64) void k_read(const d2 *src, double *sink, long long span16, long long total16) {
    const long long base = (long long)blockIdx.x * span16;
    double acc = 0;
    for (long long i = threadIdx.x; i < span16; i += 256) {
        d2 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const long long o = base + i + 64 * u; v[u] = (i + 64 * u < span16 && o < total16) ? src[o] : d2{0, 0}; }
#pragma unroll
        for (int u = 0; u < 4; u++) acc += v[u].x + v[u].y;
    }
    if (acc == 12345.678) sink[0] = acc;
}
// every line twice: block b reads span b, then span b - lag (the second touch of a span comes `lag` spans of traffic after the
// first).  Bytes requested = 2 x total; bytes that must come from DRAM = 1 x total if the re-touch is served on-die.
__global__ __launch_bounds__(64) void k_retouch(const d2 *src, double *sink, long long span16, long long total16, long long lag) {
    double acc = 0;
    for (int pass = 0; pass < 2; pass++) {
        long long blk = (long long)blockIdx.x - (pass ? lag : 0);
        if (blk < 0) blk += gridDim.x;
        const long long base = blk * span16;
        for (long long i = threadIdx.x; i < span16; i += 256) {
            d2 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const long long o = base + i + 64 * u; v[u] = (i + 64 * u < span16 && o < total16) ? src[o] : d2{0, 0}; }
#pragma unroll
            for (int u = 0; u < 4; u++) acc += v[u].x + v[u].y;
        }
    }
    if (acc == 12345.678) sink[0] = acc;
}
int main() {
    const size_t cap = (size_t)4096 << 20;
    d2 *p; double *sink;
    if (hipMalloc(&p, cap) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipMemset(p, 0, cap);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const long long span = 16384, s16 = span / 16;
    printf("== linear read of S bytes, repeated (wave span %lld B)\n", span);
    for (size_t mb : {8, 16, 24, 48, 96, 160, 224, 320, 512, 1024, 4096}) {
        const size_t bytes = mb << 20;
        const long long total16 = (long long)(bytes / 16);
        const unsigned nb = (unsigned)((total16 + s16 - 1) / s16);
        const int reps = mb <= 512 ? 40 : 8;
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k_read, dim3(nb), dim3(64), 0, 0, p, sink, s16, total16);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_read, dim3(nb), dim3(64), 0, 0, p, sink, s16, total16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("S = %5zu MiB  %9.2f us per pass  %6.2f TB/s\n", mb, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
    }
    printf("== every line twice, the second touch `lag` bytes of traffic after the first (S = 3 GiB, requested bytes = 2 S)\n");
    {
        const size_t bytes = (size_t)3072 << 20;
        const long long total16 = (long long)(bytes / 16);
        const unsigned nb = (unsigned)((total16 + s16 - 1) / s16);
        for (long long lagmb : {0LL, 1LL, 4LL, 16LL, 32LL, 64LL, 128LL, 200LL, 400LL, 1024LL}) {
            const long long lag = lagmb * (1 << 20) / span;
            for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_retouch, dim3(nb), dim3(64), 0, 0, p, sink, s16, total16, lag);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_retouch, dim3(nb), dim3(64), 0, 0, p, sink, s16, total16, lag);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("lag = %5lld MiB  %9.2f us  requested %6.2f TB/s  (unique bytes %6.2f TB/s)\n", lagmb, ms / 5 * 1e3,
                   2.0 * bytes / (ms / 5 * 1e-3) / 1e12, bytes / (ms / 5 * 1e-3) / 1e12);
        }
    }
    return 0;
}
