"""ctypes binding of include/cpi_amd.h -> cpi_amd/libcpi_amd.so (the HIP library).

There is NO CPU fallback: if the library is missing and cannot be built with hipcc this module
raises, a hipcc compile / link error propagates, a library built from other sources than the tree holds is refused
(cpi_build_id() vs build.source_id(); CPI_AMD_ALLOW_STALE=1 overrides), and creating a context on a machine without a GPU
raises (CPI_ERR_NO_DEVICE).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CPI_AMD_LIB") or os.path.join(_HERE, "libcpi_amd.so")  # env override: kernel-variant A/B runs

CPI_OK, CPI_ERR_INVALID, CPI_ERR_HIP, CPI_ERR_NO_DEVICE, CPI_ERR_RCCL = 0, 1, 2, 3, 4
ABI_VERSION = 3   # include/cpi_amd.h CPI_ABI_VERSION
OUT_FIELDS = [("DT", 1), ("alpha", 3), ("beta", 3), ("q", 4), ("J_q", 9), ("J_a", 9), ("J_b", 9),
              ("H_a", 9), ("H_b", 9), ("O_a", 9), ("O_b", 9), ("P", 225), ("P_sym", 120)]
TRI_DOUBLES = 120   # include/cpi_amd.h CPI_TRI_DOUBLES: packed upper triangle of a 15 x 15 matrix, entry (i, j), i <= j, at i + j (j + 1) / 2


class CpiParams(C.Structure):
    _fields_ = [("sigma_w", C.c_double), ("sigma_wb", C.c_double), ("sigma_a", C.c_double),
                ("sigma_ab", C.c_double), ("grav", C.c_double * 3), ("model", C.c_int32),
                ("imu_avg", C.c_int32), ("state_transition_jacobians", C.c_int32),
                ("lanes_per_window", C.c_int32)]


class CpiOutputs(C.Structure):
    _fields_ = [(name, C.c_void_p) for name, _ in OUT_FIELDS]


class CpiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("cpi_amd error %d: %s" % (code, msg))
        self.code = code


_lib = None


def _embedded_build_id(path):
    """cpi_build_id() of a library file WITHOUT loading it into this process (a stale copy must not stay mapped when the
    fresh build is loaded afterwards): the id string sits in .rodata behind a fixed tag."""
    try:
        with open(path, "rb") as f:
            blob = f.read()
        i = blob.find(b"cpi-build-id:")
        return blob[i + 13:i + 13 + 64].split(b"\0", 1)[0].decode() if i >= 0 else None
    except OSError:
        return None


def load():
    global _lib
    if _lib is not None:
        return _lib
    import shutil
    from . import build as _build
    # a machine without the compiler (detected as such, not guessed from an exception type: a missing SOURCE file is a broken
    # tree and propagates) may go on with the library that is there -- and then only when that library was built from this
    # tree's sources (cpi_build_id() == source_id(), checked below)
    hipcc_missing = None if (os.path.exists(_build.HIPCC) or shutil.which(_build.HIPCC)) else "no %s on this machine" % _build.HIPCC
    if LIB_PATH == _build.LIB and not hipcc_missing:
        # missing, or built from other sources than the tree holds; hipcc cross-compiles without a GPU.  Compile and link
        # errors PROPAGATE (a source edit that does not build must not be hidden behind the previous library).  Two notions
        # of "stale": the sidecar .id file (content hash of the sources at build time) and -- for a library copied in from
        # elsewhere beside a sidecar that says fresh -- the id compiled INTO the library, which only a forced build repairs.
        if _build.stale():
            _build.build()
        elif _embedded_build_id(LIB_PATH) != _build.source_id():
            _build.build(force=True)
    if not os.path.exists(LIB_PATH):
        raise ImportError("cpi_amd: %s is missing and could not be built (%r); no CPU fallback exists" % (LIB_PATH, hipcc_missing))
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, dp = C.c_void_p, C.c_int32, C.c_int64, C.c_void_p
    # the ABI guard comes FIRST: a stale library must fail with this message, not with an AttributeError on a new symbol
    lib.cpi_abi_version.restype = C.c_int
    if lib.cpi_abi_version() != ABI_VERSION:
        raise ImportError("cpi_amd: %s has ABI version %d, this binding expects %d (stale build?)" % (LIB_PATH, lib.cpi_abi_version(), ABI_VERSION))
    missing = [n for n in ("cpi_preintegrate_stream", "cpi_assemble_tiles", "cpi_tile_windows", "cpi_outputs_bind_slab", "cpi_preintegrate_tiled_batch_host",
                          "cpi_preintegrate_stream_host", "cpi_sqrt_information_packed_batch", "cpi_factor_eval_whitened_tri_batch",
                          "cpi_factor_hessian_tri_batch", "cpi_group_gather_chunk", "cpi_shard_chunk_bounds")
               if not hasattr(lib, n)]
    if missing:
        raise ImportError("cpi_amd: %s lacks %s (an older build: run python -m cpi_amd.build --force)" % (LIB_PATH, ", ".join(missing)))
    lib.cpi_build_id.restype = C.c_char_p
    if LIB_PATH == _build.LIB:
        # the in-tree product library must be the build of the in-tree sources: same-ABI libraries with old kernels would
        # otherwise pass every guard above.  CPI_AMD_ALLOW_STALE=1 loads it anyway (a box without hipcc and a comment-only edit).
        have, want = (lib.cpi_build_id() or b"").decode(), _build.source_id()
        if have != want:
            msg = "cpi_amd: %s was built from other sources than this tree holds (build id %s, sources %s)%s" % (
                LIB_PATH, have, want, "; hipcc is not available here (%r)" % (hipcc_missing,) if hipcc_missing else "")
            if os.environ.get("CPI_AMD_ALLOW_STALE") != "1":
                raise ImportError(msg + " -- run python -m cpi_amd.build --force, or set CPI_AMD_ALLOW_STALE=1 to load it anyway")
            import warnings
            warnings.warn(msg + " (CPI_AMD_ALLOW_STALE=1: loaded anyway)")
    lib.cpi_group_create.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]
    lib.cpi_group_destroy.argtypes = [vp]
    lib.cpi_group_destroy.restype = None
    lib.cpi_group_size.argtypes = [vp]
    lib.cpi_group_ctx.argtypes = [vp, C.c_int]
    lib.cpi_group_ctx.restype = vp
    lib.cpi_group_last_error.argtypes = [vp]
    lib.cpi_group_last_error.restype = C.c_char_p
    lib.cpi_shard_bounds.argtypes = [i64, C.c_int, C.c_int, C.POINTER(i64), C.POINTER(i64)]
    lib.cpi_shard_bounds.restype = None
    lib.cpi_group_gather.argtypes = [vp, C.c_int, i64, C.POINTER(CpiOutputs), C.POINTER(CpiOutputs)]
    lib.cpi_group_gather_chunk.argtypes = [vp, C.c_int, i64, C.c_int, C.c_int, C.POINTER(CpiOutputs), C.POINTER(CpiOutputs)]
    lib.cpi_shard_chunk_bounds.argtypes = [i64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(i64), C.POINTER(i64)]
    lib.cpi_shard_chunk_bounds.restype = None
    lib.cpi_group_synchronize.argtypes = [vp]
    lib.cpi_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    lib.cpi_ctx_destroy.argtypes = [vp]
    lib.cpi_ctx_destroy.restype = None
    lib.cpi_last_error.argtypes = [vp]
    lib.cpi_last_error.restype = C.c_char_p
    lib.cpi_ctx_synchronize.argtypes = [vp]
    lib.cpi_ctx_set_stream.argtypes = [vp, vp]
    lib.cpi_preintegrate_batch.argtypes = [vp, C.POINTER(CpiParams), i64, i32, dp, vp, vp, dp, dp, C.POINTER(CpiOutputs)]
    lib.cpi_preintegrate_tiled_batch.argtypes = [vp, C.POINTER(CpiParams), i64, i32, dp, vp, dp, dp, C.POINTER(CpiOutputs)]
    lib.cpi_tile_knots.argtypes = [vp, i64, i32, dp, dp]
    lib.cpi_stream_workspace_bytes.argtypes = [i64]
    lib.cpi_stream_workspace_bytes.restype = C.c_size_t
    lib.cpi_stream_counts.argtypes = [vp, i64]
    lib.cpi_stream_counts.restype = C.c_void_p
    lib.cpi_preintegrate_stream.argtypes = [vp, C.POINTER(CpiParams), i64, dp, i64, dp, i32, dp, dp, vp, C.POINTER(CpiOutputs)]
    lib.cpi_preintegrate_stream.restype = C.c_int
    lib.cpi_preintegrate_stream_host.argtypes = [vp, C.POINTER(CpiParams), i64, dp, i64, dp, i32, dp, dp, C.POINTER(CpiOutputs), vp]
    lib.cpi_tile_windows.argtypes = [vp, i64, i32, dp, vp, vp, dp]
    lib.cpi_assemble_tiles.argtypes = [vp, i64, dp, i64, dp, i32, dp, vp]
    lib.cpi_preintegrate_tiled_batch_host.argtypes = [vp, C.POINTER(CpiParams), i64, i32, dp, vp, dp, dp, C.POINTER(CpiOutputs)]
    lib.cpi_outputs_slab_doubles.argtypes = [C.POINTER(CpiOutputs), i64]
    lib.cpi_outputs_slab_doubles.restype = C.c_size_t
    lib.cpi_outputs_bind_slab.argtypes = [C.POINTER(CpiOutputs), i64, dp, C.POINTER(CpiOutputs)]
    lib.cpi_group_last_gather_messages.argtypes = [vp]
    if hasattr(lib, "cpi_test_group_create_shared"):   # libcpi_amd_test.so (CPI_AMD_LIB; include/cpi_amd_test.h): never the product library
        lib.cpi_test_group_create_shared.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
        lib.cpi_test_group_create_shared.restype = C.c_int
        lib.cpi_test_quat_ops.argtypes = [vp, i32, i64, dp, dp]
        lib.cpi_test_quat_ops.restype = C.c_int
    lib.cpi_factor_eval_batch.argtypes = [vp, i32, C.POINTER(C.c_double), i64, C.POINTER(CpiOutputs), dp, dp, dp, i64, vp, vp, dp, dp, dp]
    lib.cpi_factor_eval_packed_batch.argtypes = [vp, i32, C.POINTER(C.c_double), i64, C.POINTER(CpiOutputs), dp, dp, dp, i64, vp, vp, dp]
    lib.cpi_sqrt_information_batch.argtypes = [vp, i64, dp, dp]
    lib.cpi_sqrt_information_packed_batch.argtypes = [vp, i64, dp, dp]
    lib.cpi_factor_eval_whitened_tri_batch.argtypes = [vp, i32, C.POINTER(C.c_double), i64, C.POINTER(CpiOutputs), dp, dp, dp, i64, vp, vp, dp, dp, dp, dp]
    lib.cpi_factor_hessian_tri_batch.argtypes = [vp, i32, C.POINTER(C.c_double), i64, C.POINTER(CpiOutputs), dp, dp, dp, i64, vp, vp, dp, dp]
    lib.cpi_factor_eval_whitened_batch.argtypes = [vp, i32, C.POINTER(C.c_double), i64, C.POINTER(CpiOutputs), dp, dp, dp, i64, vp, vp, dp, dp, dp, dp]
    lib.cpi_factor_hessian_batch.argtypes = [vp, i32, C.POINTER(C.c_double), i64, C.POINTER(CpiOutputs), dp, dp, dp, i64, vp, vp, dp, dp]
    lib.cpi_predict_batch.argtypes = [vp, i32, C.POINTER(C.c_double), i64, C.POINTER(CpiOutputs), dp, i64, vp, dp]
    lib.cpi_preintegrate_batch_host.argtypes = [vp, C.POINTER(CpiParams), i64, i32, dp, vp, vp, i64, dp, dp, C.POINTER(CpiOutputs)]
    lib.cpi_host_alloc.argtypes = [C.c_size_t]
    lib.cpi_host_alloc.restype = C.c_void_p
    lib.cpi_host_free.argtypes = [vp]
    lib.cpi_host_free.restype = None
    lib.cpi_factor_eval_batch_host.argtypes = [vp, i32, C.POINTER(C.c_double), i64, C.POINTER(CpiOutputs), dp, dp, dp, i64, vp, vp, dp, dp, dp]
    for f in (lib.cpi_ctx_create, lib.cpi_ctx_synchronize, lib.cpi_preintegrate_batch, lib.cpi_factor_eval_batch,
              lib.cpi_sqrt_information_batch, lib.cpi_factor_eval_whitened_batch, lib.cpi_factor_eval_packed_batch,
              lib.cpi_predict_batch, lib.cpi_preintegrate_batch_host, lib.cpi_factor_eval_batch_host, lib.cpi_factor_hessian_batch,
              lib.cpi_preintegrate_tiled_batch, lib.cpi_tile_knots, lib.cpi_group_create, lib.cpi_group_gather, lib.cpi_group_synchronize, lib.cpi_group_size, lib.cpi_ctx_set_stream,
              lib.cpi_tile_windows, lib.cpi_assemble_tiles, lib.cpi_preintegrate_tiled_batch_host, lib.cpi_outputs_bind_slab,
              lib.cpi_group_last_gather_messages, lib.cpi_preintegrate_stream_host, lib.cpi_sqrt_information_packed_batch,
              lib.cpi_factor_eval_whitened_tri_batch, lib.cpi_factor_hessian_tri_batch, lib.cpi_group_gather_chunk):
        f.restype = C.c_int
    _lib = lib
    return lib
