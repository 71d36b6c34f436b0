"""CPU: the C++ facade's lazy result members (cpi_amd/csrc/cpi_host.hpp, round 5) compile with g++ against the C-ABI library and
behave without a device: fresh / stored / copied preintegrators read back exactly, and a read that needs the GPU fails loudly
(std::runtime_error carrying the library's "no HIP device" message) instead of returning stale values -- no CPU fallback."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lazy_members_without_a_device():
    import torch
    from cpi_amd import _lib
    _lib.load()
    exe = os.path.join(tempfile.mkdtemp(), "test_facade_lazy_cpu")
    libdir = os.path.join(ROOT, "cpi_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cpp", "test_facade_lazy_cpu.cpp"), "-o", exe,
                           "-L" + libdir, "-lcpi_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert p.returncode == 0, p.stdout
    if torch.cuda.is_available():
        assert p.stdout.startswith("DEVICE dt=0.005"), p.stdout
    else:
        assert p.stdout.startswith("THROWS") and "no HIP device" in p.stdout, p.stdout


def test_the_gpu_facade_program_compiles_here():
    """tests/cpp/test_facade.cpp (the GPU test's program, incl. the reference-shaped createimufactor body) must at least compile and
    link on the CPU box every round."""
    exe = os.path.join(tempfile.mkdtemp(), "test_facade")
    libdir = os.path.join(ROOT, "cpi_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", os.path.join(ROOT, "tests", "cpp", "test_facade.cpp"), "-o", exe,
                           "-L" + libdir, "-lcpi_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    assert os.path.exists(exe)
