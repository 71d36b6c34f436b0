"""CPU-only: the C-ABI shared library builds for gfx950, loads, and exports every symbol that
include/cpi_amd.h declares.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(cpi_[a-z_]+)\s*\(", src))


def _exported(lib_path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], stdout=subprocess.PIPE, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if ln.strip()}


def test_library_builds_and_exports_exactly_the_public_header():
    """The dynamic symbols of the PRODUCT library are the prototypes of include/cpi_amd.h -- all of them and nothing else: no
    test hook (round 3 shipped cpi_test_quat_ops and cpi_test_group_create_shared, the latter switches the duplicate-device
    guard off), no cpi::launch function, no std:: instantiation (cpi_amd/csrc/exports.map)."""
    from cpi_amd import _lib, build
    lib = _lib.load()
    public = _declared_symbols("cpi_amd.h")
    assert len(public) >= 20 and "cpi_group_gather" in public and not [s for s in public if s.startswith("cpi_test_")]
    for s in sorted(public):
        assert hasattr(lib, s), "libcpi_amd.so does not export %s" % s
    exported = _exported(build.LIB)
    assert exported == public, (sorted(exported - public), sorted(public - exported))
    assert lib.cpi_abi_version() == 3


def test_hooks_library_adds_exactly_the_test_header():
    """libcpi_amd_test.so = the same sources with -DCPI_TEST_HOOKS: the public ABI plus the two entries of include/cpi_amd_test.h."""
    from cpi_amd import build
    from tests import hooks_py
    path = hooks_py.lib_path()
    assert path == build.LIB_TEST and os.path.exists(path)
    hooks = _declared_symbols("cpi_amd_test.h")
    assert hooks == {"cpi_test_quat_ops", "cpi_test_group_create_shared"}
    assert _exported(path) == _declared_symbols("cpi_amd.h") | hooks


def test_struct_layouts_match_header():
    from cpi_amd._lib import CpiOutputs, CpiParams
    assert C.sizeof(CpiParams) == 7 * 8 + 4 * 4
    assert C.sizeof(CpiOutputs) == 13 * 8   # ABI 3: P_sym


def test_no_cpu_fallback():
    """Without a GPU the context constructor must fail loudly (CPI_ERR_NO_DEVICE), never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cpi_amd import _lib
    lib = _lib.load()
    ctx = C.c_void_p()
    rc = lib.cpi_ctx_create(0, None, C.byref(ctx))
    assert rc == _lib.CPI_ERR_NO_DEVICE
    assert b"no HIP device" in lib.cpi_last_error(None)
    import cpi_amd
    with pytest.raises(cpi_amd.CpiError):
        cpi_amd.Engine()


def _load_in_subprocess(prelude, env=None):
    import subprocess
    import sys
    code = "import sys, warnings; sys.path.insert(0, %r)\nfrom cpi_amd import build, _lib\n%s\nlib = _lib.load(); print('LOADED', lib.cpi_build_id().decode())" % (ROOT, prelude)
    e = dict(os.environ)
    e.pop("CPI_AMD_LIB", None)
    e.update(env or {})
    return subprocess.run([sys.executable, "-W", "always", "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)


def test_a_library_built_from_other_sources_is_refused():
    """a same-ABI library with old kernels must not pass the loader.  (i) a failing compile propagates instead of
    falling back to the library that is there; (ii) without a compiler, a library whose cpi_build_id() differs from the tree's
    source id is refused, unless CPI_AMD_ALLOW_STALE=1 says otherwise; (iii) without a compiler, the library of THIS tree loads."""
    from cpi_amd import _lib
    _lib.load()                                             # the in-tree library exists and is fresh from here on
    # (i) the tree looks edited and the compiler fails: RuntimeError from the build, not a warning
    r = _load_in_subprocess("build.stale = lambda *a, **k: True\n"
                            "def boom(*a, **k): raise RuntimeError('hipcc failed on cpi_mean.hip')\nbuild.build = boom")
    assert r.returncode != 0 and "hipcc failed on cpi_mean.hip" in r.stderr and "LOADED" not in r.stdout
    # (ii) no compiler, and the tree's sources hash differently from what the library was built from
    # (the loader asks shutil.which / os.path.exists for the compiler; build.build must then never be reached)
    no_cc = ("build.stale = lambda *a, **k: True\nbuild.HIPCC = '/nonexistent/bin/hipcc'\n"
             "def nocc(*a, **k): raise AssertionError('build() reached although there is no compiler')\nbuild.build = nocc\n")
    r = _load_in_subprocess(no_cc + "build.source_id = lambda *a, **k: '0123456789abcdef'")
    assert r.returncode != 0 and "built from other sources" in r.stderr and "LOADED" not in r.stdout
    r = _load_in_subprocess(no_cc + "build.source_id = lambda *a, **k: '0123456789abcdef'", env={"CPI_AMD_ALLOW_STALE": "1"})
    assert r.returncode == 0 and "LOADED" in r.stdout and "loaded anyway" in r.stderr
    # (iii) no compiler, sidecar lost, but the library IS this tree's build
    r = _load_in_subprocess(no_cc)
    assert r.returncode == 0 and "LOADED" in r.stdout, r.stderr
    # (iv) ADVICE round 5: the sidecar says fresh but the id compiled INTO the library differs (a .so copied in from elsewhere):
    # with a compiler present that is repaired by a FORCED build -- build() consults only the sidecar and would rebuild nothing
    r = _load_in_subprocess("build.stale = lambda *a, **k: False\n_lib._embedded_build_id = lambda p: 'feedfacefeedface'\n"
                            "real = build.build\n"
                            "def spy(*a, **k):\n    print('BUILD force=%r' % k.get('force', a[0] if a else False)); build.build = real\nbuild.build = spy")
    assert r.returncode == 0 and "BUILD force=True" in r.stdout and "LOADED" in r.stdout, (r.stdout, r.stderr)
    # a missing SOURCE file is a broken tree, not "no compiler": it propagates as what it is
    r = _load_in_subprocess("build.stale = lambda *a, **k: True\n"
                            "def gone(*a, **k): raise FileNotFoundError(2, 'No such file or directory', 'cpi_math.hpp')\nbuild.build = gone")
    assert r.returncode != 0 and "cpi_math.hpp" in r.stderr and "LOADED" not in r.stdout


def test_embedded_build_id_is_readable_without_loading():
    from cpi_amd import _lib, build
    lib = _lib.load()
    assert _lib._embedded_build_id(build.LIB) == lib.cpi_build_id().decode() == build.source_id()


def test_product_never_imports_oracle():
    """Neither the product package nor its development tools nor the C-ABI header may reference the oracle (test
    infrastructure): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do."""
    for top in ("cpi_amd", "tools", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".sh", ".hip", ".hpp", ".h", ".cpp")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "oracle_py" not in txt and "liboracle" not in txt and "cpi_oracle" not in txt, f


def test_build_id_covers_every_source_of_every_translation_unit():
    """cpi_build_id() ties measurement records to the library: every file a translation unit includes from the repository
    must be in that unit's dependency list (cpi_amd/build.py UNITS -- it keys the object cache AND feeds the hash), or an
    edit could leave a stale object / stale counters looking valid."""
    from cpi_amd import build
    hashed = {os.path.realpath(build._path(d)) for d in build._sources(experiments=True)}
    for unit, deps in build.UNITS.items():
        listed = {os.path.realpath(build._path(d)) for d in deps + build.EXP_EXTRA.get(unit, [])}
        seen, todo = set(), [os.path.join(build.CSRC, unit + ".hip")]
        while todo:
            f = os.path.realpath(todo.pop())
            if f in seen:
                continue
            seen.add(f)
            for inc in re.findall(r'^\s*#include\s+"([^"]+)"', open(f).read(), flags=re.M):
                path = os.path.realpath(os.path.join(os.path.dirname(f), inc))
                assert os.path.exists(path), (f, inc)
                todo.append(path)
        assert seen <= listed, (unit, sorted(seen - listed))
        assert seen <= hashed


def test_default_library_reads_no_environment_switches():
    """The measurement switches of tools/exp/ exist only in a -DCPI_EXPERIMENTS build: the default library's launch paths
    read no environment variable (the one getenv left is CPI_AMD_RCCL_LIB, at the first n > 1 device set)."""
    from cpi_amd import build
    src = open(os.path.join(build.CSRC, "cpi_abi.hip")).read()
    default = re.sub(r"#ifdef CPI_EXPERIMENTS.*?#endif", "", src, flags=re.S)
    assert re.findall(r'getenv\("([A-Z_]+)"\)', default) == ["CPI_AMD_RCCL_LIB"]
    for unit in ("cpi_mean", "cpi_cov", "cpi_factor"):
        assert "getenv" not in open(os.path.join(build.CSRC, unit + ".hip")).read()
    rows = open(build.REPORT).read()
    assert "cpi_mean_dma_kernel" not in rows and "cpi_mean_blk_kernel" not in rows and "probe" not in rows


def test_dpp_sources_are_not_fresh_valu_results():
    """The double-precision DPP multiply-adds (cpi_factor_kernels.hpp: dpp_fmac) are inline assembly, which hipcc's hazard
    recogniser does not inspect: "VALU write of a VGPR, then a DPP read of it: two wait states" is checked here on the
    disassembly of the shipped library (tests/tools/dpp_hazards.py), after the checker has shown it sees a planted case."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import dpp_hazards
    planted = """
0000000000001000 <k>:
	v_fma_f64 v[4:5], v[0:1], v[2:3], v[4:5]
	s_nop 0
	v_fmac_f64_dpp v[8:9], v[4:5], v[6:7] row_newbcast:3 row_mask:0xf bank_mask:0xf
	v_fma_f64 v[4:5], v[0:1], v[2:3], v[4:5]
	s_nop 1
	v_fmac_f64_dpp v[8:9], v[4:5], v[6:7] row_newbcast:3 row_mask:0xf bank_mask:0xf
	ds_read_b64 v[4:5], v0
	v_fmac_f64_dpp v[8:9], v[4:5], v[6:7] row_newbcast:3 row_mask:0xf bank_mask:0xf
	v_mov_b32_e32 v5, 0
	v_add_f64 v[10:11], v[0:1], v[2:3]
	v_mov_b64_dpp v[8:9], v[4:5] row_newbcast:1 row_mask:0xf bank_mask:0xf
"""
    n, bad = dpp_hazards.scan(planted)
    assert n == 4 and len(bad) == 2 and "s_nop 1" not in bad[0][2] and "v_mov_b32_e32 v5" in bad[1][1]
    from cpi_amd import _lib
    _lib.load()
    n, bad = dpp_hazards.check(os.path.join(ROOT, "cpi_amd", "libcpi_amd.so"))
    assert n > 500, "the Hessian sweep's DPP multiply-adds are gone?"
    assert not bad, bad[:5]


def test_three_knot_kernel_addresses_its_staged_loads_from_a_scalar_base():
    """cpi_mean_kernel<..., BIG> exists to fit three knots per chunk into two wavefronts per SIMD: ONE 32-bit offset per staged
    element against a wave-uniform base, on the fast path and on the per-element path alike.  That is a property of the generated
    code, not of the source: every staged load of the shipped kernel must have the `global_load_dwordx2 v[..], v_off32, s[base]`
    form (21 elements x {first chunk, loop} x {fast, per-element} = 84), and the kernel must fit two wavefronts per SIMD without
    scratch (cpi_amd/csrc/resource_usage.txt, written by the build)."""
    import subprocess
    import sys
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import dpp_hazards
    from cpi_amd import _lib, build
    _lib.load()
    with tempfile.TemporaryDirectory() as tmp:
        found = 0
        for co in dpp_hazards.code_objects(build.LIB, tmp):
            out = subprocess.run([os.path.join(dpp_hazards.LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co],
                                 stdout=subprocess.PIPE, text=True, check=True).stdout
            for part in re.split(r"\n(?=[0-9a-f]{16} <)", out):
                head = part.split("\n", 1)[0]
                if "cpi_mean_kernelILi1ELb0ELb0ELi1ELi0ELb1E" in head or "cpi_mean_kernelILi1ELb0ELb0ELi1ELi2ELb1E" in head:
                    loads = re.findall(r"global_load_dwordx2 v\[\d+:\d+\], (?:v\d+|v\[\d+:\d+\]), (s\[\d+:\d+\]|off)", part)
                    assert sum(1 for b in loads if b != "off") == 84, (head, len(loads))
                    found += 1
        assert found == 2
    rows = [ln.split() for ln in open(build.REPORT) if ln.startswith("cpi_mean_kernel<") and ln.split(">")[0].endswith("true")]
    assert len(rows) >= 6
    for r in rows:        # name tokens ..., SGPR VGPR AGPR scratch occ LDS
        vgpr, agpr, scratch, occ = int(r[-5]), int(r[-4]), int(r[-3]), int(r[-2])
        assert vgpr + agpr <= 256 and scratch == 0 and occ == 2, r


def test_cpp_hosts_compile_against_the_facade_without_a_gpu():
    """Every C++ host under tests/cpp/ must compile (syntax + semantics, no link, no GPU) against cpi_host.hpp and
    include/cpi_amd.h: a change of the facade or of the C-ABI that breaks a caller shows up in the CPU suite, not only on
    the GPU box.  (test_group / test_threads include the HIP runtime API for device buffers.)"""
    import subprocess
    d = os.path.join(ROOT, "tests", "cpp")
    for f in sorted(os.listdir(d)):
        if not f.endswith(".cpp"):
            continue
        cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror=return-type", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
               os.path.join(d, f)]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert p.returncode == 0, "%s:\n%s" % (f, p.stdout[-2000:])
