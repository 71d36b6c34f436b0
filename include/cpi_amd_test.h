/*
 * cpi_amd_test.h -- test hooks (NOT part of the drop-in boundary; used by tests/ only).  They exist ONLY in
 * cpi_amd/libcpi_amd_test.so -- the product sources compiled with -DCPI_TEST_HOOKS (python -m cpi_amd.build --test-hooks;
 * tests select it with CPI_AMD_LIB or tests/hooks_py.py).  The product library libcpi_amd.so exports exactly the
 * prototypes of cpi_amd.h (tests/test_abi.py).
 *
 * The device-side SO(3) / JPL-quaternion helpers of cpi_amd/csrc/cpi_math.hpp replace the reference's
 * cpi_compare/src/utils/quat_ops.h (rot_2_quat :45-86, skew_x :92-98, quat_2_Rot :104-109, quat_multiply :115-128,
 * Exp :145-162, Inv :190-197) inside every kernel; evaluateError (ImuFactorCPIv1.cpp:37-208, ImuFactorCPIv2.cpp:38-212),
 * state prediction (GraphSolver_IMU.cpp:263-307) and the preintegrators are built from them.  This entry runs each
 * helper by itself on the GPU -- the shipped device instructions (v_rsq_f64 + Newton, Horner sin/cos), not a host
 * emulation -- so that tests/test_gpu_quat_ops.py can hold them against tests/golden/quat_ops.npz, whose expected
 * values come from the reference's own functions (oracle/ref_shim.cpp: cpi_ref_quat_ops).
 */
#ifndef CPI_AMD_TEST_H
#define CPI_AMD_TEST_H

#include "cpi_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One helper call per item, one lane per item; in / out are DEVICE pointers; matrices cross this interface ROW-major.
 *   op 0 rot_2_quat    in [n][9]  -> out [n][4]      op 1 skew_x  in [n][3] -> out [n][9]      op 2 quat_2_Rot in [n][4] -> out [n][9]
 *   op 3 quat_multiply in [n][8]  -> out [n][4]      op 4 Exp     in [n][3] -> out [n][9]      op 5 Inv        in [n][4] -> out [n][4]
 * Returns CPI_OK / CPI_ERR_INVALID (unknown op, NULL pointer) / CPI_ERR_HIP. */
int cpi_test_quat_ops(cpi_ctx *ctx, int32_t op, int64_t n, const double *in, double *out);

/* A device set of n ranks that all live on ONE device: runs the n > 1 code paths of cpi_group_* (ncclCommInitAll, the
 * grouped send / recv gather, the slab unpack) on a 1-GPU box.  Real RCCL refuses duplicate devices, so the tests bind
 * tests/fake_rccl (CPI_AMD_RCCL_LIB), whose ncclSend / ncclRecv are stream-ordered device copies.  n <= 16. */
int cpi_test_group_create_shared(int n, int device, cpi_group **out);

#ifdef __cplusplus
}
#endif
#endif /* CPI_AMD_TEST_H */
