"""Development: several A/B variants of the mean translation unit at once (one hipcc per variant, in parallel), each linked with
the default build's other objects into cpi_amd/libcpi_amd_<tag>.so.

    python tools/exp/build_mean_variants.py c3w2="-DCPI_MEAN_C_L1=3 -DCPI_MEAN_WPS_L1=2" c3p="-DCPI_MEAN_C_L1=3 -DCPI_MEAN_ODD_PITCH=1"
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cpi_amd import build as b  # noqa: E402


def one(spec):
    tag, defs = spec.split("=", 1)
    defs = defs.split()
    obj = os.path.join(b.OBJ, "cpi_mean__%s.o" % tag)
    p = subprocess.run([b.HIPCC] + b.CFLAGS + defs + ["-c", "-o", obj, os.path.join(b.CSRC, "cpi_mean.hip")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    open(obj + ".log", "w").write(p.stdout)
    if p.returncode != 0:
        sys.stderr.write(p.stdout[-3000:])
        raise RuntimeError(tag)
    lib = os.path.join(b.HERE, "libcpi_amd_%s.so" % tag)
    b._link(lib, [obj] + [os.path.join(b.OBJ, u + ".o") for u in ("cpi_cov", "cpi_factor", "cpi_abi")])
    rows = [r for r in b._resource_rows(p.stdout) if "cpi_mean_kernelILi1ELb0ELb0ELi1ELi0E" in r["name"]]
    return tag, lib, rows


if __name__ == "__main__":
    b.build()
    with ThreadPoolExecutor(len(sys.argv) - 1) as ex:
        for tag, lib, rows in ex.map(one, sys.argv[1:]):
            print(tag, lib, [(r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")) for r in rows])
