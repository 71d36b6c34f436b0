// Pure-store bandwidth probes (development tool): which store pattern / cache policy reaches hipMemset's rate?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/bin/store_bw tools/exp/store_bw.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
// mode 0: plain dwordx4, a wavefront owns SPAN consecutive bytes; 1: the same with nontemporal stores;
template <int MODE>
__global__ __launch_bounds__(64) void k_span(d2 *dst, long long span16, long long total16) {
    const long long base = (long long)blockIdx.x * span16;
    for (long long i = threadIdx.x; i < span16; i += 64) {
        const long long o = base + i;
        if (o < total16) {
            d2 v = {1.0 + (double)i, 2.0};
            if (MODE == 0) dst[o] = v;
            else __builtin_nontemporal_store(v, dst + o);
        }
    }
}
// mode 2: zeros (is hipMemset's rate a property of the data?); 3: four stores in flight per lane before the loop branch
template <int MODE>
__global__ __launch_bounds__(64) void k_span2(d2 *dst, long long span16, long long total16) {
    const long long base = (long long)blockIdx.x * span16;
    if (MODE == 2) {
        for (long long i = threadIdx.x; i < span16; i += 64) if (base + i < total16) dst[base + i] = d2{0.0, 0.0};
    } else {
        for (long long i = threadIdx.x; i < span16; i += 256) {
#pragma unroll
            for (int u = 0; u < 4; u++) { const long long o = base + i + 64 * u; if (i + 64 * u < span16 && o < total16) dst[o] = d2{1.0 + (double)i, 2.0}; }
        }
    }
}
// mode 4: span per wavefront, workgroups remapped so that each XCD (blockIdx mod 8) owns one contiguous eighth; 5: 4 wavefronts per
// workgroup, consecutive spans
template <int MODE>
__global__ __launch_bounds__(256) void k_span3(d2 *dst, long long span16, long long total16, unsigned nblocks) {
    long long blk;
    if (MODE == 4) { const unsigned per = (nblocks + 7) / 8; blk = (long long)(blockIdx.x % 8) * per + blockIdx.x / 8; }
    else blk = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long base = blk * span16;
    for (long long i = threadIdx.x & 63; i < span16; i += 64) if (base + i < total16) dst[base + i] = d2{1.0 + (double)i, 2.0};
}
// reads: span per wavefront, consecutive (0) or XCD-contiguous (1) workgroups
template <int MODE>
__global__ __launch_bounds__(64) void k_read(const d2 *src, double *sink, long long span16, long long total16, unsigned nblocks) {
    long long blk = blockIdx.x;
    if (MODE == 1) { const unsigned per = (nblocks + 7) / 8; blk = (long long)(blockIdx.x % 8) * per + blockIdx.x / 8; }
    const long long base = blk * span16;
    double acc = 0;
    for (long long i = threadIdx.x; i < span16; i += 64) if (base + i < total16) { const d2 v = src[base + i]; acc += v.x + v.y; }
    if (acc == 12345.678) sink[0] = acc;
}
// mixed traffic in the proportion of the dense factor sweep: a wavefront reads rd16 pieces and writes wr16 pieces (8 factors: 6.2 KB in,
// 29.8 KB out), consecutive (0) or XCD-contiguous (1) workgroups
template <int MODE>
__global__ __launch_bounds__(64) void k_mixed(const d2 *src, d2 *dst, long long rd16, long long wr16, unsigned nblocks) {
    long long blk = blockIdx.x;
    if (MODE == 1) { const unsigned per = (nblocks + 7) / 8; blk = (long long)(blockIdx.x % 8) * per + blockIdx.x / 8; if (blk >= nblocks) return; }
    double acc = 0;
    for (long long i = threadIdx.x; i < rd16; i += 64) { const d2 v = src[blk * rd16 + i]; acc += v.x + v.y; }
    for (long long i = threadIdx.x; i < wr16; i += 64) dst[blk * wr16 + i] = d2{acc, 2.0 + (double)i};
}
// the same copy with all loads of the wavefront in flight before the first store (the sweeps' one round trip) and non-temporal stores
__global__ __launch_bounds__(64, 2) void k_mixed_nt(const d2 *src, d2 *dst, long long rd16, long long wr16, unsigned nblocks) {
    const long long blk = blockIdx.x;
    d2 v[11];
#pragma unroll
    for (int u = 0; u < 11; u++) { const long long i = threadIdx.x + 64 * u; v[u] = src[blk * rd16 + (i < rd16 ? i : rd16 - 1)]; }
    double acc = 0;
#pragma unroll
    for (int u = 0; u < 11; u++) acc += v[u].x + v[u].y;
    for (long long i = threadIdx.x; i < wr16; i += 64) __builtin_nontemporal_store(d2{acc, 2.0 + (double)i}, dst + blk * wr16 + i);
}
// grid-stride, 256-thread blocks (the classic memset shape)
template <int MODE>
__global__ __launch_bounds__(256) void k_grid(d2 *dst, long long total16) {
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total16; o += (long long)gridDim.x * 256) {
        d2 v = {1.0 + (double)o, 2.0};
        if (MODE == 0) dst[o] = v;
        else __builtin_nontemporal_store(v, dst + o);
    }
}
int main() {
    const size_t bytes = (size_t)3600 << 20;
    d2 *p;
    if (hipMalloc(&p, bytes) != hipSuccess) return 1;
    const long long total16 = (long long)(bytes / 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char *name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; r++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.3f ms  %6.2f TB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
    };
    timeit("hipMemsetAsync", [&] { hipMemsetAsync(p, 0, bytes, 0); });
    for (long long span : {1024LL, 14400LL, 65536LL, 1048576LL}) {
        const long long s16 = span / 16;
        const unsigned nb = (unsigned)((total16 + s16 - 1) / s16);
        char nm[96];
        snprintf(nm, sizeof nm, "wave span %lld B, plain", span);
        timeit(nm, [&] { hipLaunchKernelGGL(k_span<0>, dim3(nb), dim3(64), 0, 0, p, s16, total16); });
        snprintf(nm, sizeof nm, "wave span %lld B, nontemporal", span);
        timeit(nm, [&] { hipLaunchKernelGGL(k_span<1>, dim3(nb), dim3(64), 0, 0, p, s16, total16); });
    }
    for (long long span : {14400LL, 65536LL}) {
        const long long s16 = span / 16;
        const unsigned nb = (unsigned)((total16 + s16 - 1) / s16);
        char nm[96];
        snprintf(nm, sizeof nm, "wave span %lld B, zeros", span);
        timeit(nm, [&] { hipLaunchKernelGGL(k_span2<2>, dim3(nb), dim3(64), 0, 0, p, s16, total16); });
        snprintf(nm, sizeof nm, "wave span %lld B, 4 stores per trip", span);
        timeit(nm, [&] { hipLaunchKernelGGL(k_span2<3>, dim3(nb), dim3(64), 0, 0, p, s16, total16); });
    }
    for (long long span : {1024LL, 14400LL, 65536LL}) {
        const long long s16 = span / 16;
        const unsigned nb = (unsigned)((total16 + s16 - 1) / s16);
        char nm[96];
        snprintf(nm, sizeof nm, "wave span %lld B, XCD-contiguous", span);
        timeit(nm, [&] { hipLaunchKernelGGL(k_span3<4>, dim3(nb), dim3(64), 0, 0, p, s16, total16, nb); });
        snprintf(nm, sizeof nm, "wave span %lld B, 4 waves per workgroup", span);
        timeit(nm, [&] { hipLaunchKernelGGL(k_span3<5>, dim3((nb + 3) / 4), dim3(256), 0, 0, p, s16, total16, nb); });
    }
    for (long long span : {4096LL, 28672LL, 182784LL}) {
        const long long s16 = span / 16;
        const unsigned nb = (unsigned)((total16 + s16 - 1) / s16);
        char nm[96];
        snprintf(nm, sizeof nm, "READ wave span %lld B, consecutive", span);
        timeit(nm, [&] { hipLaunchKernelGGL(k_read<0>, dim3(nb), dim3(64), 0, 0, p, (double *)p, s16, total16, nb); });
        snprintf(nm, sizeof nm, "READ wave span %lld B, XCD-contiguous", span);
        timeit(nm, [&] { hipLaunchKernelGGL(k_read<1>, dim3(nb), dim3(64), 0, 0, p, (double *)p, s16, total16, nb); });
    }
    {
        const long long rd16 = 6208 / 16, wr16 = 29760 / 16;
        const unsigned nb = 125000;
        d2 *src;
        if (hipMalloc(&src, (size_t)nb * rd16 * 16) != hipSuccess) return 1;
        hipMemset(src, 0, (size_t)nb * rd16 * 16);
        const double gb = (double)nb * (rd16 + wr16) * 16;
        auto t2 = [&](const char *name, auto launch) {
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; r++) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-44s %8.3f ms  %6.2f TB/s (read + write)\n", name, ms / 5, gb / (ms / 5 * 1e-3) / 1e12);
        };
        t2("MIXED 6.2 KB in / 29.8 KB out, consecutive", [&] { hipLaunchKernelGGL(k_mixed<0>, dim3(nb), dim3(64), 0, 0, src, p, rd16, wr16, nb); });
        t2("MIXED 6.2 KB in / 29.8 KB out, XCD-contig.", [&] { hipLaunchKernelGGL(k_mixed<1>, dim3((nb + 7) / 8 * 8), dim3(64), 0, 0, src, p, rd16, wr16, nb); });
    }
    // round 5: the mixes of the whitened sweep (4 factors per wavefront: 4 x (776 + 1 800) B in, 4 x 3 720 B out) and of the Hessian
    // sweep (4 x 2 576 in, 4 x 3 968 out), 200 000 wavefronts -- is 0.57-0.62 of 8 TB/s the ceiling of a 1 : 1.5 read : write copy?
    for (int which = 0; which < 2; which++) {
        const long long rd16 = 10304 / 16, wr16 = (which == 0 ? 14880 : 15872) / 16;
        const unsigned nb = 200000;                       // 200 000 x 15 872 B = 3.17 GB of the 3 600-MiB destination
        d2 *src;
        if (hipMalloc(&src, (size_t)nb * rd16 * 16) != hipSuccess) return 1;
        hipMemset(src, 0, (size_t)nb * rd16 * 16);
        const double gb = (double)nb * (rd16 + wr16) * 16;
        auto t3 = [&](const char *name, auto launch) {
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; r++) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-52s %8.3f ms  %6.2f TB/s (read + write) = %.3f of 8 TB/s\n", name, ms / 5, gb / (ms / 5 * 1e-3) / 1e12, gb / (ms / 5 * 1e-3) / 8e12);
        };
        t3(which == 0 ? "MIXED whitened sweep 10.3 KB in / 14.9 KB out" : "MIXED Hessian sweep 10.3 KB in / 15.9 KB out",
           [&] { hipLaunchKernelGGL(k_mixed<0>, dim3(nb), dim3(64), 0, 0, src, p, rd16, wr16, nb); });
        t3(which == 0 ? "  the same, non-temporal stores" : "  the same, non-temporal stores",
           [&] { hipLaunchKernelGGL(k_mixed_nt, dim3(nb), dim3(64), 0, 0, src, p, rd16, wr16, nb); });
        hipFree(src);
    }
    timeit("hipMemsetAsync 0xff", [&] { hipMemsetAsync(p, 0xff, bytes, 0); });
    for (unsigned nb : {65536u}) {
        char nm[96];
        snprintf(nm, sizeof nm, "grid-stride %u x 256, plain", nb);
        timeit(nm, [&] { hipLaunchKernelGGL(k_grid<0>, dim3(nb), dim3(256), 0, 0, p, total16); });
        snprintf(nm, sizeof nm, "grid-stride %u x 256, nontemporal", nb);
        timeit(nm, [&] { hipLaunchKernelGGL(k_grid<1>, dim3(nb), dim3(256), 0, 0, p, total16); });
    }
    return 0;
}
