// cpi_factor.hip -- translation unit of the re-linearisation sweeps: evaluateError (dense / packed / whitened / Hessian
// blocks; ImuFactorCPIv1.cpp:37-208, ImuFactorCPIv2.cpp:38-212), square-root information (ImuFactorCPIv1.h:82), state
// prediction (GraphSolver_IMU.cpp:263-307), plus two small utility kernels (the slab unpack of cpi_group_gather and the
// test hook of include/cpi_amd_test.h), with their launchers (cpi_args.hpp: cpi::launch).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "cpi_args.hpp"
#include "cpi_math.hpp"

using namespace cpi;

#include "cpi_device_util.hpp"
#include "cpi_factor_kernels.hpp"

namespace {
// Root side of cpi_group_gather's slab path: rank r's slab (all wanted fields of its block, field-major over wb_r windows)
// sits in staging at r * stride; field k of its cnt_r windows goes to root_out.k + lo_r * n_k.  One launch for all ranks
// and fields; a workgroup column copies one rank's pieces, consecutive threads = consecutive doubles.
struct UnpackArgs {
    int n;
    long long lo[16], cnt[16], wb[16];
    const double *staging;
    long long stride;
    cpi_outputs out;
};
__global__ __launch_bounds__(256) void cpi_unpack_slabs_kernel(UnpackArgs U) {
    const int NF[13] = { 1, 3, 3, 4, 9, 9, 9, 9, 9, 9, 9, 225, CPI_TRI_DOUBLES };
    double *const dst[13] = { U.out.DT, U.out.alpha, U.out.beta, U.out.q, U.out.J_q, U.out.J_a, U.out.J_b, U.out.H_a,
                              U.out.H_b, U.out.O_a, U.out.O_b, U.out.P, U.out.P_sym };
    const int r = blockIdx.y;
    if (U.cnt[r] <= 0) return;
    long long off = 0;
#pragma unroll
    for (int k = 0; k < 13; k++) {
        if (!dst[k]) continue;
        const long long len = U.cnt[r] * NF[k];
        const double *s = U.staging + (long long)r * U.stride + off;
        double *d = dst[k] + U.lo[r] * NF[k];
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < len; i += (long long)gridDim.x * 256) d[i] = s[i];
        off += U.wb[r] * NF[k];
    }
}

#ifdef CPI_TEST_HOOKS   // include/cpi_amd_test.h: libcpi_amd_test.so only
__global__ __launch_bounds__(64) void cpi_test_quat_ops_kernel(int op, long long n, const double *in, double *out) {
    const long long k = (long long)blockIdx.x * 64 + threadIdx.x;
    if (k >= n) return;
    auto put_rm = [&](double *o, const M3 &A) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) o[i * 3 + j] = A.m[i][j];
    };
    auto put_q = [&](double *o, Q4 q) { o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w; };
    switch (op) {
        case 0: put_q(out + 4 * k, rot_2_quat(rec_mat(in + 9 * k, 0))); break;
        case 1: put_rm(out + 9 * k, skew(ldv3(in + 3 * k))); break;
        case 2: put_rm(out + 9 * k, quat_2_Rot(ldq4(in + 4 * k))); break;
        case 3: put_q(out + 4 * k, quat_multiply(ldq4(in + 8 * k), ldq4(in + 8 * k + 4))); break;
        case 4: put_rm(out + 9 * k, Exp_so3(ldv3(in + 3 * k))); break;
        default: put_q(out + 4 * k, quat_inv(ldq4(in + 4 * k))); break;
    }
}
#endif
}  // namespace

namespace cpi {
namespace launch {

void factor(int model, bool whiten, int lpf, const FactorArgs &a, hipStream_t st) {
    const long long F = a.F;
#define CPI_LAUNCH_FACTOR(M, WH, L, TR) \
    hipLaunchKernelGGL((cpi_factor_kernel<M, WH, L, TR>), dim3(factor_grid((F + 64 / L - 1) / (64 / L))), dim3(64), 0, st, a)
#define CPI_LAUNCH_FACTOR_L(M, WH, TR) \
    do { if (lpf == 16) CPI_LAUNCH_FACTOR(M, WH, 16, TR); else if (lpf == 8) CPI_LAUNCH_FACTOR(M, WH, 8, TR); else CPI_LAUNCH_FACTOR(M, WH, 4, TR); } while (0)
    if (whiten && a.r_tri) {
        if (model == CPI_MODEL_V1) CPI_LAUNCH_FACTOR_L(1, true, true); else CPI_LAUNCH_FACTOR_L(2, true, true);
    } else if (whiten) {
        if (model == CPI_MODEL_V1) CPI_LAUNCH_FACTOR_L(1, true, false); else CPI_LAUNCH_FACTOR_L(2, true, false);
    } else {
        if (model == CPI_MODEL_V1) CPI_LAUNCH_FACTOR_L(1, false, false); else CPI_LAUNCH_FACTOR_L(2, false, false);
    }
#undef CPI_LAUNCH_FACTOR_L
#undef CPI_LAUNCH_FACTOR
}

void factor_packed(int model, int lpf, const FactorArgs &a, double *packed, hipStream_t st) {
    const long long F = a.F;
#define CPI_PACKED(M, L) hipLaunchKernelGGL((cpi_factor_packed_kernel<M, L>), dim3(factor_grid((F + 64 / L - 1) / (64 / L))), dim3(64), 0, st, a, packed)
    if (model == CPI_MODEL_V1) { if (lpf == 2) CPI_PACKED(1, 2); else if (lpf == 3) CPI_PACKED(1, 3); else if (lpf == 4) CPI_PACKED(1, 4); else if (lpf == 6) CPI_PACKED(1, 6); else CPI_PACKED(1, 8); }
    else                       { if (lpf == 2) CPI_PACKED(2, 2); else if (lpf == 3) CPI_PACKED(2, 3); else if (lpf == 4) CPI_PACKED(2, 4); else if (lpf == 6) CPI_PACKED(2, 6); else CPI_PACKED(2, 8); }
#undef CPI_PACKED
}

void factor_hessian(int model, const FactorArgs &a, double *hess, hipStream_t st) {
    const unsigned nb = factor_grid((a.F + 3) / 4);
    if (a.r_tri) {
        if (model == CPI_MODEL_V1) hipLaunchKernelGGL((cpi_factor_hessian_kernel<1, true>), dim3(nb), dim3(64), 0, st, a, hess);
        else hipLaunchKernelGGL((cpi_factor_hessian_kernel<2, true>), dim3(nb), dim3(64), 0, st, a, hess);
    } else {
        if (model == CPI_MODEL_V1) hipLaunchKernelGGL((cpi_factor_hessian_kernel<1, false>), dim3(nb), dim3(64), 0, st, a, hess);
        else hipLaunchKernelGGL((cpi_factor_hessian_kernel<2, false>), dim3(nb), dim3(64), 0, st, a, hess);
    }
}

void sqrt_info(long long F, const double *P, double *R, bool packed, hipStream_t st) {
    const long long groups = (F + 3) / 4;
    const unsigned nb = CPI_SQRT_WPB > 1 ? (unsigned)((groups + CPI_SQRT_WPB - 1) / CPI_SQRT_WPB) : factor_grid(groups);
    if (packed) hipLaunchKernelGGL(cpi_sqrt_info_kernel<true>, dim3(nb), dim3(64 * CPI_SQRT_WPB), 0, st, F, P, R);
    else hipLaunchKernelGGL(cpi_sqrt_info_kernel<false>, dim3(nb), dim3(64 * CPI_SQRT_WPB), 0, st, F, P, R);
}

void predict(int model, const PredictArgs &a, hipStream_t st) {
    const long long nb = (a.F + 63) / 64;
    if (model == CPI_MODEL_V1) hipLaunchKernelGGL((cpi_predict_kernel<1>), dim3((unsigned)nb), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((cpi_predict_kernel<2>), dim3((unsigned)nb), dim3(64), 0, st, a);
}

#ifdef CPI_TEST_HOOKS
void test_quat_ops(int op, long long n, const double *in, double *out, hipStream_t st) {
    hipLaunchKernelGGL(cpi_test_quat_ops_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, op, n, in, out);
}
#endif

void unpack_slabs(int n, const long long *lo, const long long *cnt, const long long *wb, const double *staging, long long stride,
                  const cpi_outputs &root_out, hipStream_t st) {
    UnpackArgs U;
    U.n = n;
    long long most = 0;
    for (int r = 0; r < 16; r++) {
        U.lo[r] = r < n ? lo[r] : 0; U.cnt[r] = r < n ? cnt[r] : 0; U.wb[r] = r < n ? wb[r] : 0;
        if (U.cnt[r] > most) most = U.cnt[r];
    }
    U.staging = staging; U.stride = stride; U.out = root_out;
    if (most <= 0) return;
    const unsigned bx = (unsigned)std::min<long long>(std::max<long long>((most * 225 + 255) / 256, 1), 4096);
    hipLaunchKernelGGL(cpi_unpack_slabs_kernel, dim3(bx, (unsigned)n), dim3(256), 0, st, U);
}

}  // namespace launch
}  // namespace cpi
