"""GPU, >= 2 devices: the first runs of the N > 1 path on REAL RCCL.  Skipped on the pool's 1-GPU boxes (where
tests/test_gpu_group.py runs the same code through the stand-in and a gloo rehearsal); live the moment a multi-GPU node runs
the suite -- they VALIDATE (bitwise against the unsharded call) instead of merely timing:

  * the single-process device set of the C-ABI (cpi_group_create -> ncclCommInitAll on distinct devices, cpi_group_gather:
    one ncclSend per peer straight to the root) -- tests/tools/group_real_check.py;
  * `python bench.py --gpus N` (N = 2, 4 and every device of the node) with no rehearsal switch: one rank per GPU over ProcessGroupNCCL, `config.rccl` says what the
    collective library saw, rank 0 recomputes every rank's last-step batch and compares the gathered blocks bitwise
    (`config.gather_verified`), and the line separates kernel time from the collective tail (`value_kernel_only`)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NDEV = torch.cuda.device_count() if torch.cuda.is_available() else 0
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(NDEV < 2, reason="needs >= 2 GPUs (real RCCL refuses two ranks on one device)")]


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CPI_BENCH_SINGLE_DEVICE", "CPI_AMD_RCCL_LIB", "CPI_AMD_LIB", "CPI_BENCH_FORCE_DIST"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["CPI_BENCH_STRICT"] = "1"          # a gathered block that differs from rank 0's recomputation is a failure, not a field
    env["NCCL_DEBUG"] = "VERSION"          # the collective library says which build it is (one line at communicator creation)
    return env


NS = sorted(x for x in {2, 4, NDEV} if 2 <= x <= max(NDEV, 2))     # an 8-GPU node also checks the intermediate points of the curve


@pytest.mark.parametrize("n", NS)
def test_device_set_on_real_rccl_reproduces_the_unsharded_call_bitwise(n):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "group_real_check.py"), str(n)], env=_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0 and "group_real_check ok" in p.stdout, p.stdout[-3000:]


@pytest.mark.parametrize("n", NS)
@pytest.mark.parametrize("extra", [[], ["--workload", "v2_full", "--windows", "20000", "--scaling", "strong"], ["--workload", "v1_full", "--windows", "30000"],
                                   ["--workload", "cfg5_mean", "--windows", "200000"],
                                   # round 6: the slab with the packed covariance (ABI 3), and the exchange inside one batch
                                   ["--workload", "v1_full_sym", "--windows", "30000"],
                                   ["--workload", "cfg5_full_sym", "--windows", "100000", "--gather-schedule", "chunked", "--gather-chunks", "8"],
                                   ["--workload", "cfg5_full", "--windows", "100000", "--gather-schedule", "chunked", "--gather-chunks", "1"]])
def test_bench_gpus_n_on_real_rccl_validates_what_it_gathers(n, extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "20", "--warmup", "5"] + extra,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.strip().splitlines() if ln.strip()][-1]
    assert len(line) < 6000
    d = json.loads(line)
    assert d["n_gpus"] == n and d["value"] > 0 and d["steps"] == 20 and "REHEARSAL" not in d["data"] and d["data"] == "synthetic"
    c = d["config"]
    rc = c["rccl"]
    assert rc["backend"] == "nccl" and rc["world_size"] == n and rc["distinct_devices"] == n and rc["nccl_version"]
    assert sorted(r[0] for r in rc["ranks"]) == list(range(n)) and sorted(r[1] for r in rc["ranks"]) == list(range(n))
    assert c["gather_verified"] is True and "bitwise" in c["gather_verified_how"]
    assert c["value_kernel_only"] >= d["value"] and c["value_without_gather"] > 0
    assert c["kernel_ms"] > 0 and c["gather_ms"] > 0 and c["wall_ms"] >= c["kernel_ms"] * 0.99
    assert d["scaling"] == ("strong" if "strong" in extra else "weak") and "timed region" in c["scaling_note"]
    # round 6: the library's banner and the node's link types travel with the record; the prediction stands beside the measurement
    assert "version" in (p.stdout + p.stderr).lower()
    assert isinstance(rc["link_types"], (dict, str)) and rc["link_types"]
    pr = c["predicted"]
    assert pr["peers"] == n - 1 and pr["exchange_ms_per_slab"] > 0 and pr["expected_value"] > 0
    if "chunked" in extra:
        assert c["gather_schedule"] == "chunked" and "sub-block" in c["launch_mode"]
