"""cpi_amd -- MI355X-native batched IMU continuous preintegration (CPI) engine.

The product is the HIP library cpi_amd/libcpi_amd.so behind the C-ABI of include/cpi_amd.h;
this package is the thin host-side mirror of the reference's CpiV1/CpiV2 and ImuFactorCPIv1/v2
interfaces over that ABI.  There is no CPU fallback.
"""
from ._lib import CpiError, LIB_PATH  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not need torch / a GPU
    if name in ("Engine", "CpiV1", "CpiV2", "ForsterDiscrete", "ImuFactorCPIv1", "ImuFactorCPIv2", "default_engine", "unpack_factor",
                "pack_sym", "pack_tri", "unpack_sym", "unpack_tri"):
        from . import engine
        return getattr(engine, name)
    if name == "EnginePool":
        from .pool import EnginePool
        return EnginePool
    raise AttributeError(name)
