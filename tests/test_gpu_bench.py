"""GPU: the DRIVER's bench command, verbatim (`python bench.py --gpus 1 --steps 20 --warmup 5`, no shortening flags): the
LAST stdout line is one JSON object small enough for the driver to keep whole (< 6 KB; round 3's line had grown to 33 KB and
`BENCH_r03.parsed` came back null), carries the contract's keys, the roofline of the dominant kernel, the CPU baseline, a
compact BASELINE configs[2] object, and names the file that holds the full extra rows."""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER_ARGS = ["--gpus", "1", "--steps", "20", "--warmup", "5"]


@pytest.fixture(scope="module")
def driver_run():
    extra_path = os.path.join(ROOT, "bench_extra.json")
    if os.path.exists(extra_path):
        os.remove(extra_path)
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + DRIVER_ARGS, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=1500, cwd=ROOT)
    wall = time.perf_counter() - t0
    assert p.returncode == 0, p.stderr[-3000:]
    return p.stdout, wall, extra_path


def test_the_last_stdout_line_of_the_drivers_command_parses_and_is_small(driver_run):
    stdout, wall, _ = driver_run
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    line = lines[-1]
    assert len(line) < 6000, len(line)
    # what the driver keeps is a tail of stdout: everything it needs must be inside the last 8 000 characters
    assert stdout.rstrip("\n")[-8000:].endswith(line)
    r = json.loads(line)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert r["metric"].startswith(base["metric"].split(" at ")[0]) and r["unit"] == "windows/s"
    assert r["n_gpus"] == 1 and r["steps"] == 20 and r["warmup"] == 5
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["dtype"] == "f64" and r["data"] == "synthetic" and "model" not in r["config"]
    for k in ("workload", "library_build", "launch_mode"):
        assert r["config"][k], k
    assert "configs[1]" in r["config"]["workload"]
    assert r["value"] > 1e8 and abs(r["value"] * r["ms_per_step"] * 1e-3 / 10000 - 1.0) < 1e-6      # 10 k windows per step
    assert r["ms_per_step"] * 1e-3 * r["steps"] <= wall                                              # fits in the run that produced it
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["kernel"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.05 < rf["frac"] < 1.0
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["launch_us"] * 1e-6) / 1e9) < 1e-6 * rf["achieved"]
    assert rf["launch_us"] * 1e-3 <= r["ms_per_step"] * 1.001                                      # kernel time <= step time
    assert rf["traffic"] is None or rf["traffic"] > 0.9 * rf["algorithmic_bytes_per_launch"]
    cb = r["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "windows/s"
    assert cb["single_core_value"] > 0 and 0 < len(cb["sample"]) <= 200 and cb["sparse_port"]["value"] > 0
    # the north-star's 40 % criterion is answered in the line itself
    assert r["goal_40pct_hbm"] == (rf["frac"] >= 0.40) and "17 k" in r["goal_note"]
    # ... and so is the one route to it at this batch size: several batches in flight, MEASURED in the same run (round 5)
    ov = r["overlapped"]
    assert ov["contexts"] == 3 and ov["unit"] == "windows/s" and ov["value"] > r["value"] * 0.8 and ov["batches"] >= 1000
    assert abs(ov["value"] * ov["ms_per_batch"] * 1e-3 / 10000 - 1.0) < 1e-9
    assert abs(ov["frac"] - ov["value"] * rf["algorithmic_bytes_per_unit"] / 8e12) < 1e-9 and 0.1 < ov["frac"] < 1.0
    assert set(ov["frac_by_contexts"]) == {"2", "3", "4"} and r["goal_40pct_hbm_overlapped"] == (ov["frac"] >= 0.40)
    # BASELINE configs[2] (V2 + covariance + Jacobians: the row the >= 10 M windows/s goal sits on) travels in the parsed part
    c2 = r["configs2"]
    assert "configs[2]" in c2["workload"] and c2["value"] > 1e7 and c2["goal_10M_windows_per_s"] is True and c2["launch_ms"] > 0
    assert c2["cpu_baseline"]["value"] > 0 and c2["cpu_baseline"]["cores"] >= 1 and c2["cpu_baseline"]["kind"] in ("reference", "port")
    assert set(c2["fp64"]) >= {"frac", "useful_frac"}
    # `value` is the lightest configuration: the full integrator's rate and the overlapped rate stand beside it at the top level
    assert r["value_full_integrator"] == c2["value"] and r["value_overlapped"] == ov["value"]
    rt = r["routes_1M_x_50"]
    assert rt["stream_in_place"] > 0 and rt["assemble_tiles"] > 0 and rt["tiled_kernel"] > 0 and rt["dense_kernel_preassembled"] > 0
    assert "bench_extra.json" in r["extra_file"]


def test_the_full_extra_rows_land_in_bench_extra_json(driver_run):
    stdout, _, extra_path = driver_run
    r = json.loads([ln for ln in stdout.splitlines() if ln.strip()][-1])
    doc = json.load(open(extra_path))
    rows = doc["rows"]
    assert len(rows) >= 45
    assert not [x for x in rows if "error" in x], [x for x in rows if "error" in x]
    with_roof = [x for x in rows if "roofline" in x]
    assert len(with_roof) >= 44
    for x in with_roof:
        rf = x["roofline"]
        assert rf["achieved"] > 0 and rf["kernel"] and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert doc["headline"]["value"] == r["value"] and doc["headline"]["config"]["library_build"] == r["config"]["library_build"]
    # every measured row is pre-ramped like the headline (late round 6: a row that starts on an idle GPU timed the power controller's transient)
    import bench
    assert all(x.get("clock_preramp_ms") == bench.EXTRA_PRERAMP_MS for x in with_roof) and r["config"]["clock_preramp_ms"] > 0
    # the compact per-row summary of the line agrees with the file
    key = bench.row_key
    assert set(r["extra_rows"]) == {key(x) for x in rows}
    by = {key(x): x for x in rows}
    # round 6: the reference's own window lengths (10 / 20 samples) with roofline AND the reference's CPU leg at that length
    for k in ("v1_mean@1Mx10", "v1_mean@1Mx20", "v1_full@1Mx10", "v2_full@1Mx20"):
        assert by[k]["roofline"]["frac"] > 0 and by[k]["cpu_baseline"]["value"] > 0 and "%d-sample" % by[k]["samples"] in by[k]["cpu_baseline"]["sample"], k
    assert by["v1_mean@1Mx20"]["roofline"]["frac"] >= 0.50
    # ... and the packed-triangle rows of ABI 3 beside their dense twins
    assert by["sqrt_info_packed@1M"]["launch_ms"] < 0.70 * by["sqrt_info@1M"]["launch_ms"]
    # (rows of one run are measured minutes apart at whatever clock the box holds then: the same-box alternating A/B of the two
    #  forms is profiles/r06_packed.md; here only "not slower beyond the noise")
    assert by["factor_v1_whitened_tri@1M"]["launch_ms"] < 1.05 * by["factor_v1_whitened@1M"]["launch_ms"]


def test_no_extra_no_cpu_still_prints_one_contract_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "50", "--warmup", "10", "--no-extra", "--no-cpu"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["steps"] == 50 and "configs2" not in r and "cpu_baseline" not in r and "overlapped" not in r and r["roofline"]["frac"] > 0.05
