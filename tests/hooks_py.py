"""Loader for cpi_amd/libcpi_amd_test.so -- the library built with -DCPI_TEST_HOOKS, i.e. the product sources plus the two
entries of include/cpi_amd_test.h (cpi_test_quat_ops, cpi_test_group_create_shared).  TEST INFRASTRUCTURE ONLY: the product
library (cpi_amd/libcpi_amd.so) exports nothing outside include/cpi_amd.h (tests/test_abi.py).

    lib_path()   builds the library when stale; subprocesses select it with CPI_AMD_LIB (cpi_amd/_lib.py)
    lib()        an in-process ctypes handle BESIDE the product library (RTLD_LOCAL: its symbols bind to itself)
"""
import ctypes as C

_lib = None


def lib_path():
    from cpi_amd import build
    return build.build_test_hooks()


def lib():
    global _lib
    if _lib is None:
        h = C.CDLL(lib_path())
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        h.cpi_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
        h.cpi_ctx_destroy.argtypes = [vp]
        h.cpi_ctx_destroy.restype = None
        h.cpi_ctx_synchronize.argtypes = [vp]
        h.cpi_last_error.argtypes = [vp]
        h.cpi_last_error.restype = C.c_char_p
        h.cpi_test_quat_ops.argtypes = [vp, i32, i64, vp, vp]
        h.cpi_test_group_create_shared.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
        for f in (h.cpi_ctx_create, h.cpi_ctx_synchronize, h.cpi_test_quat_ops, h.cpi_test_group_create_shared):
            f.restype = C.c_int
        _lib = h
    return _lib
