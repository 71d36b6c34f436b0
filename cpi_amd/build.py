"""Build libcpi_amd.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

    python -m cpi_amd.build                      # build what is stale
    python -m cpi_amd.build --force --report     # rebuild everything and print the per-kernel resource table
    python -m cpi_amd.build --experiments        # additionally libcpi_amd_exp.so (-DCPI_EXPERIMENTS: the measurement-only
                                                 # kernels and environment switches of tools/exp/; load it with CPI_AMD_LIB)
    python -m cpi_amd.build --test-hooks         # additionally libcpi_amd_test.so (-DCPI_TEST_HOOKS: the two entries of
                                                 # include/cpi_amd_test.h; the PRODUCT library exports nothing but include/cpi_amd.h)

The library is four translation units (cpi_amd/csrc/cpi_args.hpp) compiled IN PARALLEL into cpi_amd/csrc/_obj/*.o and
linked into one shared object; an object is rebuilt only when the sources it includes (or the flags) change, so touching
one kernel family costs one TU.  hipcc cross-compiles gfx950 without a GPU present; the .so is git-ignored but ships to
the GPU box.
"""
import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libcpi_amd.so")
LIB_EXP = os.path.join(HERE, "libcpi_amd_exp.so")
LIB_TEST = os.path.join(HERE, "libcpi_amd_test.so")
REPORT = os.path.join(CSRC, "resource_usage.txt")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Rpass-analysis=kernel-resource-usage"]
COMMON = ["cpi_args.hpp", "../../include/cpi_amd.h", "exports.map"]   # exports.map: the link step's version script
DEVICE = COMMON + ["cpi_math.hpp", "cpi_device_util.hpp"]
# translation unit -> the files it includes (what its object depends on)
UNITS = {
    "cpi_mean": DEVICE + ["cpi_mean.hip", "cpi_mean_kernels.hpp"],
    "cpi_cov": DEVICE + ["cpi_cov.hip", "cpi_cov_kernels.hpp"],
    "cpi_factor": DEVICE + ["cpi_factor.hip", "cpi_factor_kernels.hpp"],
    "cpi_abi": COMMON + ["cpi_abi.hip", "../../include/cpi_amd_test.h"],   # the test header: -DCPI_TEST_HOOKS builds only
}
EXP_EXTRA = {"cpi_mean": ["cpi_mean_experimental.hpp"]}   # additional includes under -DCPI_EXPERIMENTS
EXP_UNITS = ("cpi_mean", "cpi_abi")                        # the units that differ in an experiments build
# library variants: path, define, the units compiled with the define (the other objects are shared with the default build),
# object-name suffix.  "exp" = measurement kernels / environment switches, "test" = the hooks of include/cpi_amd_test.h.
VARIANTS = {
    "": (LIB, None, (), ""),
    "exp": (LIB_EXP, "-DCPI_EXPERIMENTS", EXP_UNITS, "_exp"),
    "test": (LIB_TEST, "-DCPI_TEST_HOOKS", ("cpi_factor", "cpi_abi"), "_test"),
}


def _path(rel):
    return os.path.normpath(os.path.join(CSRC, rel))


def _sources(experiments=False):
    s = set()
    for u, deps in UNITS.items():
        s.update(deps)
        if experiments:
            s.update(EXP_EXTRA.get(u, []))
    return sorted(s)


def source_id(experiments=False, variant=None):
    """sha256[:16] over the sources the library is built from -- compiled into it (cpi_build_id()), so that measurement
    records (profiles/*_pmc.json) can be tied to the exact library that is loaded."""
    variant = ("exp" if experiments else "") if variant is None else variant
    h = hashlib.sha256()
    for d in _sources(variant == "exp"):
        with open(_path(d), "rb") as f:
            h.update(d.encode() + b"\0" + f.read())
    if variant == "exp":
        h.update(b"CPI_EXPERIMENTS")
    elif variant:
        h.update(VARIANTS[variant][1].encode())
    return h.hexdigest()[:16]


def _unit_key(unit, defines, experiments):
    h = hashlib.sha256(" ".join(CFLAGS + defines).encode())
    for d in UNITS[unit] + (EXP_EXTRA.get(unit, []) if experiments else []):
        with open(_path(d), "rb") as f:
            h.update(d.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def _compile(unit, variant, force):
    _, define, vunits, suffix = VARIANTS[variant]
    own = bool(variant) and unit in vunits          # this unit differs from the default build's object
    exp = own and variant == "exp"
    defines = ([define] if own else [])
    if unit == "cpi_abi":
        defines.append('-DCPI_BUILD_ID="%s"' % source_id(variant=variant))
    obj = os.path.join(OBJ, unit + (suffix if own else "") + ".o")
    stamp, log = obj + ".key", obj + ".log"
    key = _unit_key(unit, defines, exp)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == key:
        return obj, (open(log).read() if os.path.exists(log) else "")
    cmd = [HIPCC] + CFLAGS + defines + ["-c", "-o", obj, os.path.join(CSRC, unit + ".hip")]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stdout)
        raise RuntimeError("hipcc failed on %s.hip" % unit)
    with open(log, "w") as f:
        f.write(p.stdout)
    with open(stamp, "w") as f:
        f.write(key)
    return obj, p.stdout


def _resource_rows(text):
    rows, cur = [], {}
    for line in text.splitlines():
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]):\s*(\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        else:
            cur[k.split(" ")[0]] = v
    return rows


def _link(lib, objs):
    """Linked under a temporary name and moved into place: a process that loads the library while another one rebuilds it
    sees the old file or the new one, never a half-written one."""
    tmp = "%s.tmp.%d" % (lib, os.getpid())
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"),
           "-o", tmp] + objs + ["-ldl"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stdout)
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("link failed: %s" % lib)
    os.replace(tmp, lib)


class _BuildLock:
    """One builder at a time per tree (several ranks / pytest subprocesses may import the package at once and would
    otherwise write the same _obj/*.o and .so concurrently).  Advisory flock on a file under _obj/."""

    def __enter__(self):
        import fcntl
        os.makedirs(OBJ, exist_ok=True)
        self.f = open(os.path.join(OBJ, ".lock"), "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *a):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()
        return False


def stale(lib=LIB, experiments=False, variant=None):
    """Content-based (a sidecar file holds the source id the library was built from): file times do not survive a copy of
    the tree to another machine, and a rebuild there would burn GPU-box minutes for nothing."""
    try:
        with open(lib + ".id") as f:
            return (not os.path.exists(lib)) or f.read().strip() != source_id(experiments, variant)
    except OSError:
        return True


def build(force=False, report=False, experiments=False, test_hooks=False):
    """Returns the path of the default library (with experiments / test_hooks the additional variants are built too:
    libcpi_amd_exp.so, libcpi_amd_test.so)."""
    with _BuildLock():
        return _build_locked(force, report, experiments, test_hooks)


def build_test_hooks():
    """libcpi_amd_test.so, built when stale; returns its path (tests/: CPI_AMD_LIB for subprocesses, tests/hooks_py.py in-process)."""
    build(test_hooks=True)
    return LIB_TEST


def build_custom(tag, defines):
    """A/B builds of kernel variants (development; tools/exp/): every unit compiled with the extra defines into
    _obj/<unit>__<tag>.o and linked into cpi_amd/libcpi_amd_<tag>.so -- load it with CPI_AMD_LIB.  Never the product path.
        python -m cpi_amd.build --custom lean0 -DCPI_MEAN_LEAN=0"""
    lib = os.path.join(HERE, "libcpi_amd_%s.so" % tag)
    with _BuildLock():
        def one(unit):
            # a unit is rebuilt only when one of the files it includes mentions one of the macros; the others are the default build's objects
            exp = "-DCPI_EXPERIMENTS" in defines          # an experiments variant: the unit also includes its EXP_EXTRA headers
            text = "".join(open(_path(d), errors="ignore").read() for d in UNITS[unit] + (EXP_EXTRA.get(unit, []) if exp else []))
            defs = [d for d in defines if d[2:].split("=")[0] in text]
            if unit == "cpi_abi":
                defs.append('-DCPI_BUILD_ID="%s"' % (source_id() + "+" + tag)[:31])
            elif not defs:
                return _compile(unit, "", False)[0]
            obj = os.path.join(OBJ, "%s__%s.o" % (unit, tag))
            key = _unit_key(unit, defs, exp)
            if os.path.exists(obj) and os.path.exists(obj + ".key") and open(obj + ".key").read() == key:
                return obj
            p = subprocess.run([HIPCC] + CFLAGS + defs + ["-c", "-o", obj, os.path.join(CSRC, unit + ".hip")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if p.returncode != 0:
                sys.stderr.write(p.stdout)
                raise RuntimeError("hipcc failed on %s.hip (%s)" % (unit, tag))
            with open(obj + ".log", "w") as f:
                f.write(p.stdout)
            with open(obj + ".key", "w") as f:
                f.write(key)
            return obj
        with ThreadPoolExecutor(len(UNITS)) as ex:
            objs = list(ex.map(one, UNITS))
        _link(lib, objs)
    return lib


def _build_locked(force, report, experiments, test_hooks):
    todo = []   # staleness is judged INSIDE the lock: whoever waited for another builder finds the library fresh
    for variant, wanted in (("", True), ("exp", experiments), ("test", test_hooks)):
        if wanted and (force or stale(VARIANTS[variant][0], variant=variant)):
            todo.append(variant)
    # every object of every wanted variant in ONE pool (the default build's four units + the units a variant compiles with its own
    # define): __graft_entry__.build() rebuilds three libraries in the time of the slowest translation unit
    jobs = {}
    for variant in todo:
        vunits = VARIANTS[variant][2]
        for u in UNITS:
            own = bool(variant) and u in vunits
            jobs.setdefault((u, variant if own else ""), force and (not variant or own))
    with ThreadPoolExecutor(min(len(jobs), os.cpu_count() or 4) or 1) as ex:
        done = dict(zip(jobs, ex.map(lambda k: _compile(k[0], k[1], jobs[k]), jobs)))
    for variant in todo:
        lib, _, vunits, _ = VARIANTS[variant]
        exp = variant
        res = [done[(u, variant if (variant and u in vunits) else "")] for u in UNITS]
        _link(lib, [o for o, _ in res])
        with open(lib + ".id", "w") as f:
            f.write(source_id(variant=variant))
        if not exp:
            rows = [r for _, text in res for r in _resource_rows(text)]
            with open(REPORT, "w") as f:
                f.write("%-72s %5s %5s %5s %7s %4s %6s\n" % ("kernel", "SGPR", "VGPR", "AGPR", "scratch", "occ", "LDS"))
                names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), stdout=subprocess.PIPE, text=True).stdout.splitlines()
                for r, name in zip(rows, names):
                    name = name.strip().replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                    f.write("%-72s %5s %5s %5s %7s %4s %6s\n" % (name, r.get("TotalSGPRs"), r.get("VGPRs"), r.get("AGPRs"),
                                                             r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
    return LIB


if __name__ == "__main__":
    if "--custom" in sys.argv:
        i = sys.argv.index("--custom")
        print(build_custom(sys.argv[i + 1], [a for a in sys.argv[i + 2:] if a.startswith("-D")]))
        sys.exit(0)
    build(force="--force" in sys.argv, report="--report" in sys.argv, experiments="--experiments" in sys.argv,
          test_hooks="--test-hooks" in sys.argv)
    print(LIB)
    if "--report" in sys.argv:
        print(open(REPORT).read())
