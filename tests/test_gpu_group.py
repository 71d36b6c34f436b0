"""Device-set entries of the C-ABI (cpi_group_*, include/cpi_amd.h).  The test box has ONE GPU, so:
  * a set of one device runs natively (block partition, per-rank contexts / streams, the root's own block landing at its
    offset; RCCL is not even loaded for n = 1; n > the number of devices is refused);
  * the n > 1 code -- ncclCommInitAll, the grouped send / recv gather, the one-slab-per-peer path with its unpack launch,
    short and empty trailing ranks -- runs with 2..16 ranks that SHARE the device (include/cpi_amd_test.h:
    cpi_test_group_create_shared) over tests/fake_rccl, a stand-in whose ncclSend / ncclRecv are stream-ordered device
    copies, and must reproduce the unsharded call bit for bit (tests/tools/group_check.py, own process);
  * a librccl that cannot be bound is CPI_ERR_RCCL with a readable message, not a crash.
The torch.distributed twin of the exchange is covered on CPU by tests/test_dist_gloo.py."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_n_rank_device_sets_on_one_device_reproduce_the_unsharded_call_bitwise():
    from tests import fake_rccl_py, hooks_py
    env = dict(os.environ, CPI_AMD_RCCL_LIB=fake_rccl_py.lib_path(), CPI_AMD_LIB=hooks_py.lib_path())   # the library WITH the hooks
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "group_check.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0 and "group_check ok" in p.stdout, p.stdout[-3000:]


def test_unloadable_rccl_is_an_error_code_not_a_crash():
    """the dlopen failure path appended a NULL dlerror() to a std::string.  Own process: the binding is made
    once per process."""
    code = ("import ctypes as C, sys; sys.path.insert(0, %r)\n"
            "from cpi_amd import _lib\nlib = _lib.load()\ng = C.c_void_p()\n"
            "rc = lib.cpi_test_group_create_shared(2, 0, C.byref(g))\n"
            "msg = lib.cpi_group_last_error(None).decode()\n"
            "assert rc == _lib.CPI_ERR_RCCL and not g, rc\n"
            "assert 'dlopen(/nonexistent/librccl.so.1)' in msg and len(msg) > 40, msg\n"
            "rc = lib.cpi_test_group_create_shared(2, 0, C.byref(g))\n"       # and again: the failed state is not sticky garbage
            "assert rc == _lib.CPI_ERR_RCCL\nprint('rccl-load ok:', msg)\n") % ROOT
    from tests import hooks_py
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CPI_AMD_RCCL_LIB="/nonexistent/librccl.so.1", CPI_AMD_LIB=hooks_py.lib_path()),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0 and "rccl-load ok" in p.stdout, p.stdout[-2000:]


def test_shard_bounds_match_the_python_partition():
    from cpi_amd import _lib
    from cpi_amd.dist import shard_bounds
    lib = _lib.load()
    for W in (0, 1, 7, 8, 9, 10000, 8000001):
        for n in (1, 2, 3, 8):
            for r in range(n):
                lo, hi = C.c_int64(), C.c_int64()
                lib.cpi_shard_bounds(W, r, n, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == shard_bounds(W, r, n)[:2]


def test_group_of_one_device_runs_the_hot_path_and_gathers():
    import cpi_amd
    from cpi_amd import _lib, synth
    from cpi_amd._lib import CpiOutputs
    lib = _lib.load()
    g = C.c_void_p()
    assert lib.cpi_group_create(1, None, C.byref(g)) == 0, lib.cpi_group_last_error(None)
    try:
        assert lib.cpi_group_size(g) == 1
        ctx = lib.cpi_group_ctx(g, 0)
        assert ctx and lib.cpi_group_ctx(g, 1) is None
        dev = torch.device("cuda", 0)
        W, N = 1000, 50
        kn, lin, q = synth.make_windows(W, N, seed=8, device=dev)
        eng = cpi_amd.Engine(device=0)
        ref = eng.preintegrate(kn, lin, q, eng.make_params(1))
        torch.cuda.synchronize()
        # the group's context (own stream) computes the block, the gather entry places it in the root arrays
        loc = eng.alloc_outputs(W, ("mean", "jac", "cov"), 1)
        root = {k: torch.full_like(v, float("nan")) for k, v in loc.items()}
        prm = eng.make_params(1)
        o = eng._outputs_struct(loc)
        torch.cuda.synchronize()
        assert lib.cpi_preintegrate_batch(ctx, C.byref(prm), W, N, kn.data_ptr(), None, None, lin.data_ptr(), q.data_ptr(), C.byref(o)) == 0
        locs = (CpiOutputs * 1)(o)
        ro = eng._outputs_struct(root)
        assert lib.cpi_group_gather(g, 0, W, locs, C.byref(ro)) == 0, lib.cpi_group_last_error(g)
        assert lib.cpi_group_synchronize(g) == 0
        for k in ref:
            assert torch.equal(root[k], ref[k]), k
        # a field wanted at the root but missing locally is an error, not a silent skip
        o2 = eng._outputs_struct({k: v for k, v in loc.items() if k != "P"})
        assert lib.cpi_group_gather(g, 0, W, (CpiOutputs * 1)(o2), C.byref(ro)) == _lib.CPI_ERR_INVALID
    finally:
        lib.cpi_group_destroy(g)


def test_group_larger_than_the_machine_is_refused():
    from cpi_amd import _lib
    lib = _lib.load()
    g = C.c_void_p()
    n = torch.cuda.device_count() + 1
    assert lib.cpi_group_create(n, None, C.byref(g)) == _lib.CPI_ERR_INVALID and not g
    assert b"number of devices" in lib.cpi_group_last_error(None)
    dup = (C.c_int * 2)(0, 0)
    if torch.cuda.device_count() >= 2:
        assert lib.cpi_group_create(2, dup, C.byref(g)) == _lib.CPI_ERR_INVALID


def test_factor_indices_are_clamped_on_the_device_and_validated_on_the_host_path():
    """ABI 2: the factor entries know the number of states.  Out-of-range indices must not read out of bounds (device
    path: clamped) and are rejected by the host-pointer variant."""
    import cpi_amd
    from cpi_amd import _lib, synth
    eng = cpi_amd.Engine(device=0)
    F = 64
    kn, lin, q = synth.make_windows(F, 20, seed=2, device=eng.device)
    meas = eng.preintegrate(kn, lin, q, eng.make_params(1), want=("mean", "jac"))
    xi, xj = synth.make_states(meas["alpha"], meas["beta"], meas["q"], meas["DT"], lin, 1, device=eng.device)
    states = torch.cat([xi, xj[-1:]], dim=0).contiguous()          # S = F + 1
    S = states.shape[0]
    good = eng.factor_eval(1, meas, lin, None, states)
    ii = torch.arange(F, dtype=torch.int32, device=eng.device)
    jj = ii + 1
    bad_j = jj.clone(); bad_j[-1] = 2 ** 30                         # far outside: clamped to S - 1 = the right state here
    bad_i = ii.clone(); bad_i[0] = -5                               # clamped to 0 = the right state here
    out = eng.factor_eval(1, meas, lin, None, states, bad_i, bad_j)
    torch.cuda.synchronize()
    for k in ("err", "H1", "H2"):
        assert torch.equal(out[k], good[k]), k
    with pytest.raises(cpi_amd.CpiError):                           # chained (NULL) indices need S >= F + 1
        eng.factor_eval(1, meas, lin, None, states[:F].contiguous())
    # host-pointer variant: validates
    lib = _lib.load()
    m = {k: v.cpu().numpy() for k, v in meas.items()}
    ms = eng._outputs_struct({k: torch.from_numpy(v) for k, v in m.items()})
    linh, sth = lin.cpu().numpy(), states.cpu().numpy()
    err = np.zeros((F, 15))
    bi = bad_i.cpu().numpy()
    gv = (C.c_double * 3)(0.0, 0.0, 9.8)
    rc = lib.cpi_factor_eval_batch_host(eng.ctx, 1, gv, F, C.byref(ms), linh.ctypes.data, None, sth.ctypes.data, S,
                                        bi.ctypes.data, None, err.ctypes.data, None, None)
    assert rc == _lib.CPI_ERR_INVALID and b"out of range" in lib.cpi_last_error(eng.ctx)


def test_cpp_host_sharding_a_batch_over_the_device_set(golden_dir):
    """tests/cpp/test_group.cpp: a single-process C++ host (g++, HIP runtime API only) shards the golden batch over the
    node's GPUs through cpi_host::DeviceGroup, gathers on GPU 0 and must reproduce the compiled reference's golden
    outputs.  On the 1-GPU test box the set has one device; on an 8-GPU node the same binary runs 8 ranks + RCCL."""
    import os
    import subprocess
    import tempfile
    from tests.tol import check_pre
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from cpi_amd import _lib
    _lib.load()
    tmp = tempfile.mkdtemp()
    exe, exe_hooks = os.path.join(tmp, "test_group"), os.path.join(tmp, "test_group_hooks")
    libdir = os.path.join(ROOT, "cpi_amd")
    from tests import hooks_py
    hooks_py.lib_path()
    base = ["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "cpp", "test_group.cpp")]
    tail = ["-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(base + ["-o", exe, "-L" + libdir, "-lcpi_amd"] + tail)                                   # the product library
    subprocess.check_call(base + ["-DCPI_TEST_HOOKS", "-o", exe_hooks, "-L" + libdir, "-l:libcpi_amd_test.so"] + tail)   # + "shared" mode
    d = dict(np.load(os.path.join(golden_dir, "pre_w48.npz")))
    kn, lin, q = d["knots"], d["lin"], d["q_k_lin"]
    W, n1, _ = kn.shape
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        np.array([W, n1 - 1], dtype=np.float64).tofile(f)
        kn.tofile(f); lin.tofile(f); q.tofile(f)
        path = f.name
    for model in (1, 2):
        p = subprocess.run([exe, path, str(model)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        rows = np.array([[float(x) for x in ln.split()] for ln in p.stdout.strip().split("\n")])
        assert rows.shape == (W, 236)
        got = {"DT": rows[:, 0], "alpha": rows[:, 1:4], "beta": rows[:, 4:7], "q": rows[:, 7:11], "P": rows[:, 11:]}
        key = "m%d_avg0_stj1__" % model
        ref = {k[len(key):]: v for k, v in d.items() if k.startswith(key)}
        check_pre(got, ref, what=("mean", "cov"), regression=True)
        assert "gathered on rank 0" in p.stderr
        # the same binary with 5 ranks sharing the device over the RCCL stand-in: 48 windows -> blocks of 10, 10, 10, 10, 8,
        # one slab message per peer, identical output text (same kernels on the same windows)
        from tests import fake_rccl_py
        p5 = subprocess.run([exe_hooks, path, str(model), "5", "shared"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                            env=dict(os.environ, CPI_AMD_RCCL_LIB=fake_rccl_py.lib_path()))
        assert p5.returncode == 0, p5.stderr
        assert "group of 5 rank(s) on one device" in p5.stderr and "1 message(s) per peer" in p5.stderr, p5.stderr
        assert p5.stdout == p.stdout
        # round 6: the exchange INSIDE the batch through cpi_host::DeviceGroup::gather_chunk -- 3 sub-blocks per rank, the slabs carry
        # the packed covariance (P_sym), the host unpacks it: identical text again; one device, and 5 ranks over the stand-in
        pc = subprocess.run([exe, path, str(model), "1", "native", "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert pc.returncode == 0 and pc.stdout == p.stdout and "in 3 sub-blocks" in pc.stderr, pc.stderr
        pc5 = subprocess.run([exe_hooks, path, str(model), "5", "shared", "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                             env=dict(os.environ, CPI_AMD_RCCL_LIB=fake_rccl_py.lib_path()))
        assert pc5.returncode == 0 and pc5.stdout == p.stdout, pc5.stderr
        assert "group of 5 rank(s) on one device" in pc5.stderr and "in 3 sub-blocks, 3 message(s) per peer in all" in pc5.stderr, pc5.stderr


@pytest.mark.parametrize("n", [4, 8])
def test_bench_gpus_4_and_8_the_drivers_first_multi_gpu_commands_rehearsed_on_one_gpu(n):
    """The exact commands the driver issues for its scaling curve -- `bench.py --gpus N --steps 20 --warmup 5`, N = 4 and 8, all
    defaults -- as a rehearsal on this 1-GPU box (CPI_BENCH_SINGLE_DEVICE=1: N ranks compute on cuda:0, the exchange goes through
    gloo).  What must hold the first time real RCCL runs it holds here: the line parses, is < 6 KB, says n_gpus = N, every rank's
    last-step slab arrives bitwise at rank 0 (CPI_BENCH_STRICT), the N pools of resident batches fit beside each other, the line
    says how long the timed region is against the exchange (`scaling_note`), and the per-GPU share of BASELINE configs[4]
    (`--workload cfg5_mean`: 1 M windows x 100 samples per rank) runs the pipelined schedule with N ranks' slabs verified."""
    import json
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["CPI_BENCH_SINGLE_DEVICE"] = "1"
    env["CPI_BENCH_STRICT"] = "1"
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    runs = [(["--gpus", str(n), "--steps", "20", "--warmup", "5"], 20, 10000, "final")]
    if n == 8:
        runs.append((["--gpus", str(n), "--steps", "3", "--warmup", "1", "--workload", "cfg5_mean", "--windows", "200000"], 3, 200000, "pipelined"))
    if n == 4:
        # round 6: the exchange INSIDE one batch (the torch.distributed twin of cpi_group_gather_chunk) on the full-V1 share of
        # configs[4], with the covariance as its packed upper triangle in the slab, 1 and 8 sub-blocks per step
        for k in (1, 8):
            runs.append((["--gpus", str(n), "--steps", "2", "--warmup", "1", "--workload", "cfg5_full_sym", "--windows", "60000",
                          "--gather-schedule", "chunked", "--gather-chunks", str(k)], 2, 60000, "chunked"))
    exposed = {}
    for args, steps, W, sched in runs:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           timeout=1200, env=env, cwd=ROOT)
        assert p.returncode == 0, "\n".join(ln for ln in p.stderr.splitlines() if "Traceback" in ln or "Error" in ln or "bench.py" in ln or "assert" in ln)[-3000:]
        lines = [ln for ln in p.stdout.strip().splitlines() if ln.strip()]
        line = lines[-1]
        assert line.startswith("{") and len(line) < 6000, (len(line), line[:200])
        d = json.loads(line)
        assert d["n_gpus"] == n and d["steps"] == steps and d["value"] > 0 and d["scaling"] == "weak" and "REHEARSAL" in d["data"]
        assert abs(d["value"] * d["ms_per_step"] * 1e-3 / (n * W) - 1.0) < 1e-6          # whole-job aggregate: N ranks x W windows per step
        c = d["config"]
        assert c["rccl"]["backend"] == "gloo" and c["rccl"]["world_size"] == n and sorted(r[0] for r in c["rccl"]["ranks"]) == list(range(n))
        assert c["gather_schedule"] == sched and c["gather_verified"] is True and "bitwise" in c["gather_verified_how"]
        assert c["value_kernel_only"] >= d["value"] and c["value_without_gather"] > 0
        assert "timed region" in c["scaling_note"] and "exchange" in c["scaling_note"] and len(c["scaling_note"]) < 900
        assert "roofline" in d and "cpu_baseline" not in d and "overlapped" not in d       # rank 0 at N = 1 only
        # round 6: the line states what the exchange SHOULD cost on xGMI before anyone measures it
        pr = c["predicted"]
        fields_doubles = {"final": 11, "pipelined": 11, "chunked": 176}[sched]           # mean-only slab / full V1 with packed P
        assert pr["slab_doubles_per_window"] == fields_doubles and pr["slab_bytes_per_peer"] == W * fields_doubles * 8 and pr["peers"] == n - 1
        assert pr["link_GBs_one_way"] == 76.8 and abs(pr["exchange_ms_per_slab"] - pr["slab_bytes_per_peer"] / 76.8e9 * 1e3) < 1e-9
        assert pr["schedule"] == sched and pr["expected_value"] > 0 and pr["expected_value"] <= pr["expected_value_without_gather"] * (1 + 1e-9)
        assert 0 < pr["kernel_ms_per_step_measured"] * steps <= c["wall_ms"] * 1.5      # the K steps without the exchange
        if sched == "chunked":
            assert "sub-block" in c["launch_mode"] and pr["chunks"] == int(args[-1])
            exposed[int(args[-1])] = c["gather_ms"]
    if n == 4:
        # both sub-block counts ran, were verified bitwise, and reported their exposed exchange tail (gloo through host memory: the
        # numbers say nothing about xGMI -- the schedule and its verification are what this exercises)
        assert set(exposed) == {1, 8} and all(v > 0 for v in exposed.values())


def test_cpp_host_threads_one_context_each(golden_dir):
    """tests/cpp/test_threads.cpp: four host threads, one context each, run the host-pointer preintegration (pipelined
    staging owned by the context) and the factor sweep concurrently; each must match the single-threaded run bit for bit."""
    import os
    import subprocess
    import tempfile
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from cpi_amd import _lib, synth
    _lib.load()
    exe = os.path.join(tempfile.mkdtemp(), "test_threads")
    libdir = os.path.join(ROOT, "cpi_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "test_threads.cpp"), "-o", exe, "-L" + libdir, "-lcpi_amd",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    W, N = 70001, 10                                         # two pipeline chunks per call
    kn, lin, q = synth.make_windows(W, N, seed=77)
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        np.array([W, N], dtype=np.float64).tofile(f)
        kn.numpy().tofile(f); lin.numpy().tofile(f); q.numpy().tofile(f)
        path = f.name
    p = subprocess.run([exe, path, "4"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0 and "threads ok T=4" in p.stdout, (p.returncode, p.stdout, p.stderr)


def test_bench_gpus_2_end_to_end_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher: re-executes under torch.distributed.run with two ranks, shards, runs
    the kernels, gathers the last step's slabs to rank 0, reduces the timing over the ranks and prints ONE line with
    n_gpus = 2.  On this 1-GPU box both ranks compute on cuda:0 and the exchange goes through gloo
    (CPI_BENCH_SINGLE_DEVICE=1: RCCL refuses two ranks on one device) -- a rehearsal of the control flow, not a measurement."""
    import json
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["CPI_BENCH_SINGLE_DEVICE"] = "1"
    env["CPI_BENCH_STRICT"] = "1"
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    for extra in ([], ["--workload", "v2_full", "--windows", "20000", "--scaling", "strong"]):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-extra",
                            "--no-cpu"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env, cwd=ROOT)
        assert p.returncode == 0, "\n".join(ln for ln in p.stderr.splitlines() if "Traceback" in ln or "Error" in ln or "bench.py" in ln or "assert" in ln)[-3000:]
        line = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")][-1]
        d = json.loads(line)
        assert d["n_gpus"] == 2 and d["value"] > 0 and d["steps"] == 5
        c = d["config"]
        assert "value_without_gather" in c and c["value_without_gather"] >= d["value"] * 0.5
        assert d["scaling"] == ("strong" if extra else "weak")
        assert "REHEARSAL" in d["data"]
        # the exchange is timed on its own, the schedule is named: one final gather for the 10 k-window headline, every
        # step's slab overlapped with the next step for the covariance row
        assert c["gather_schedule"] == ("pipelined" if extra else "final") and c["kernel_ms"] > 0 and c["gather_ms"] > 0
        assert c["wall_ms"] >= c["kernel_ms"] * 0.99 and abs(d["ms_per_step"] * 5 - c["wall_ms"]) < 1e-6 * c["wall_ms"]
        assert ("graph" in c["launch_mode"]) == (not extra)
        # round 4: the line says what the process group saw (here: gloo, two ranks on ONE device -- which is why it is a
        # rehearsal), separates the kernels from the collective tail, and rank 0 has recomputed every rank's last-step batch
        # and compared the gathered blocks bitwise
        assert c["rccl"]["backend"] == "gloo" and c["rccl"]["world_size"] == 2 and c["rccl"]["distinct_devices"] == 1
        assert c["value_kernel_only"] >= d["value"]
        assert c["gather_verified"] is True and "bitwise" in c["gather_verified_how"]
        assert len(line) < 6000
