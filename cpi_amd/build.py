"""Build libcpi_amd.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

    python -m cpi_amd.build            # build if stale
    python -m cpi_amd.build --force --report   # rebuild and write the per-kernel resource table

hipcc cross-compiles gfx950 without a GPU present; the .so is git-ignored but ships to the GPU box.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "cpi_kernels.hip")
KERNEL_PARTS = ["cpi_device_util.hpp", "cpi_mean_kernels.hpp", "cpi_mean_experimental.hpp", "cpi_cov_kernels.hpp", "cpi_factor_kernels.hpp"]
DEPS = ([SRC, os.path.join(HERE, "csrc", "cpi_math.hpp")] + [os.path.join(HERE, "csrc", f) for f in KERNEL_PARTS] +
        [os.path.join(os.path.dirname(HERE), "include", "cpi_amd.h")])
LIB = os.path.join(HERE, "libcpi_amd.so")
REPORT = os.path.join(HERE, "csrc", "resource_usage.txt")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast",
         "-Rpass-analysis=kernel-resource-usage"]


def source_id():
    """sha256[:16] over the sources the library is built from -- compiled into it (cpi_build_id()), so that measurement
    records (profiles/*_pmc.json) can be tied to the exact library that is loaded."""
    import hashlib
    h = hashlib.sha256()
    for d in DEPS + [os.path.join(os.path.dirname(HERE), "include", "cpi_amd_test.h")]:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def stale():
    return (not os.path.exists(LIB)) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS)


def build(force=False, report=False):
    if not (force or stale()):
        return LIB
    cmd = [HIPCC] + FLAGS + ['-DCPI_BUILD_ID="%s"' % source_id(), "-o", LIB, SRC]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stdout)
        raise RuntimeError("hipcc failed")
    rows, cur = [], {}
    for line in p.stdout.splitlines():
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]):\s*(\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        else:
            cur[k.split(" ")[0]] = v
    if report or True:
        with open(REPORT, "w") as f:
            f.write("%-72s %5s %5s %5s %7s %4s %6s\n" % ("kernel", "SGPR", "VGPR", "AGPR", "scratch", "occ", "LDS"))
            for r in rows:
                name = subprocess.run(["c++filt", r["name"]], stdout=subprocess.PIPE, text=True).stdout.strip()
                name = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                f.write("%-72s %5s %5s %5s %7s %4s %6s\n" % (name, r.get("TotalSGPRs"), r.get("VGPRs"), r.get("AGPRs"),
                                                         r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, report="--report" in sys.argv)
    print(LIB)
    if "--report" in sys.argv:
        print(open(REPORT).read())
