// What does the memory system deliver for the dense layout's access pattern?  (development tool, round 5: profiles/r05_mean_traffic.md)
// A wavefront owns 64 windows of STRIDE bytes each (dense layout: 2856) and visits them in lock step: per visit it reads B
// contiguous bytes of every window (LW = 8 or 16 bytes per lane and load, 64 LW bytes per load instruction, 8 KB per visit kept in
// flight), then moves on by B bytes -- the staged mean kernels' fetch with nothing else in the way.  Windows of a wavefront:
// consecutive (GAP = 1: the staged kernels) or every 16th (GAP = 16: the phase-sorted line kernel).  B = 128 ... 2816.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/bin/stride_probe tools/exp/stride_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int LW, int OCC>
__global__ __launch_bounds__(64, OCC) void k_visit(const char *src, double *sink, long long W, int stride, int B, int gap, int visits) {
    // windows of this wavefront: gap == 1: 64 b + i; gap == 16: 1024 (b / 16) + (b % 16) + 16 i
    const long long b = blockIdx.x;
    const long long w0 = (gap == 1) ? b * 64 : (b / gap) * 64 * gap + (b % gap);
    const int lane = threadIdx.x;
    const int lpw = B / LW;                    // lanes per window of one load instruction (B <= 64 LW) or 64
    const int per = lpw >= 64 ? 1 : 64 / lpw;  // windows per load instruction
    const int ni = (B * 64) / (64 * LW);       // load instructions per visit = B / LW ... capped below
    double acc = 0;
    for (int v = 0; v < visits; ++v) {
        const long long vb = (long long)v * B;
        if (lpw <= 64) {
            // instruction e covers windows per e .. per e + per - 1
            for (int e0 = 0; e0 * per < 64; e0 += 8) {
                double t[8][2];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u;
                    const long long win = w0 + (long long)gap * min(per * e + lane / lpw, 63);
                    const char *p = src + win * stride + vb + (long long)(lane % lpw) * LW;
                    if (LW == 16) { const d2 x = *reinterpret_cast<const d2 *>(p); t[u][0] = x.x; t[u][1] = x.y; }
                    else { t[u][0] = *reinterpret_cast<const double *>(p); t[u][1] = 0; }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += t[u][0] + t[u][1];
            }
        }
        (void)ni;
    }
    if (acc == 12345.678) sink[0] = acc;
}
template <int LW, int OCC>
static float run(const char *p, double *sink, long long W, int stride, int B, int gap) {
    const int visits = stride / B;             // whole visits only (the tail of a window is not read: bytes counted accordingly)
    const unsigned nb = (unsigned)(W / 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_visit<LW, OCC>), dim3(nb), dim3(64), 0, 0, p, sink, W, stride, B, gap, visits);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k_visit<LW, OCC>), dim3(nb), dim3(64), 0, 0, p, sink, W, stride, B, gap, visits);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}
int main() {
    const long long W = 1 << 20;               // windows
    const int stride = 2856;
    char *p; double *sink;
    if (hipMalloc(&p, (size_t)W * stride + 4096) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    hipMemset(p, 0, (size_t)W * stride + 4096);
    printf("64 windows of %d B per wavefront, %lld windows; B bytes per window and visit, LW bytes per lane and load\n", stride, W);
    for (int gap : {1, 16}) for (int LW : {8, 16}) for (int B : {112, 128, 168, 176, 224, 256, 512, 1024}) {
        if (B / LW > 64 || B % LW != 0) continue;
        const int visits = stride / B;
        const int lpw = B / LW, per = lpw >= 64 ? 1 : 64 / lpw;
        (void)per;                                             // (window indices are clamped to the wavefront's 64: all of them are read)
        const double bytes = (double)W * visits * B;
        float ms2 = (LW == 8) ? run<8, 2>(p, sink, W, stride, B, gap) : run<16, 2>(p, sink, W, stride, B, gap);
        float ms3 = (LW == 8) ? run<8, 3>(p, sink, W, stride, B, gap) : run<16, 3>(p, sink, W, stride, B, gap);
        printf("gap %2d  LW %2d  B %4d: %8.1f us %5.2f TB/s (2 waves/SIMD)   %8.1f us %5.2f TB/s (3 waves/SIMD)\n", gap, LW, B,
               ms2 * 1e3, bytes / (ms2 * 1e-3) / 1e12, ms3 * 1e3, bytes / (ms3 * 1e-3) / 1e12);
    }
    return 0;
}
