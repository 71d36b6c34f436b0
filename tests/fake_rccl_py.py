"""Loader for tests/fake_rccl/libfake_rccl.so -- the test-only RCCL stand-in (see fake_rccl.cpp).  TEST INFRASTRUCTURE ONLY."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "fake_rccl", "fake_rccl.cpp")
LIB = os.path.join(_HERE, "fake_rccl", "libfake_rccl.so")


def lib_path():
    if (not os.path.exists(LIB)) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                               "-o", LIB, SRC, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    return LIB
