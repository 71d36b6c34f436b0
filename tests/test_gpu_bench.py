"""GPU: the driver's bench command end to end -- one JSON line carrying the contract's keys, the roofline object of the dominant
kernel and the CPU baseline of the same workload, internally consistent."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_shape():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "50", "--warmup", "10", "--no-extra"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert r["metric"].startswith(base["metric"].split(" at ")[0]) and r["unit"] == "windows/s"
    assert r["n_gpus"] == 1 and r["steps"] == 50 and r["warmup"] == 10
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["dtype"] == "f64" and r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    assert r["value"] > 1e8 and abs(r["value"] * r["ms_per_step"] * 1e-3 / 10000 - 1.0) < 1e-6      # 10 k windows per step
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.05 < rf["frac"] < 1.0
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["launch_us"] * 1e-6) / 1e9) < 1e-6 * rf["achieved"]
    assert rf["launch_us"] * 1e-3 <= r["ms_per_step"] * 1.001                                      # kernel time <= step time
    assert rf["traffic"] is None or rf["traffic"] > 0.9 * rf["algorithmic_bytes_per_launch"]
    cb = r["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "windows/s" and cb["sample"]
