// cpi_cov.hip -- translation unit of the covariance (+ state transition) kernels: cpi_cov_kernel<1|2> (CpiV1.h:266-353,
// CpiV2.h:314-464) and cpi_forster_kernel (the GTSAM discrete comparator, GraphSolver_IMU.cpp:141-232), with their
// launchers (cpi_args.hpp: cpi::launch).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cpi_args.hpp"
#include "cpi_math.hpp"

using namespace cpi;

#include "cpi_device_util.hpp"
#include "cpi_cov_kernels.hpp"

namespace cpi {
namespace launch {

template <int MODEL>
static void launch_cov(bool avg, const PreArgs &a, hipStream_t st) {
    constexpr int G = 64 / CovDims<MODEL>::GROUP;
    const long long nb = (a.W + G - 1) / G;
    if (avg) hipLaunchKernelGGL((cpi_cov_kernel<MODEL, true>), dim3((unsigned)nb), dim3(64), 0, st, a);
    else     hipLaunchKernelGGL((cpi_cov_kernel<MODEL, false>), dim3((unsigned)nb), dim3(64), 0, st, a);
}
void cov(int model, bool avg, const PreArgs &a, hipStream_t st) {
    if (model == CPI_MODEL_V2) launch_cov<2>(avg, a, st); else launch_cov<1>(avg, a, st);
}
void forster(const PreArgs &a, hipStream_t st) {   // one kernel owns everything; imu_avg, q_k_lin, grav play no part
    hipLaunchKernelGGL(cpi_forster_kernel, dim3((unsigned)((a.W + 3) / 4)), dim3(64), 0, st, a);
}

}  // namespace launch
}  // namespace cpi
