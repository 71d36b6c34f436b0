"""Kernel A/B harness (development tool): time several (workload, windows, lanes) configs in one process.
   CPI_AMD_LIB=build/exp/variant.so python tools/microbench.py v1_mean:10000:8 v1_mean:1000000:1 ...
Prints launch microseconds (HIP events over `steps` back-to-back launches on a >256 MiB batch pool)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import cpi_amd  # noqa: E402


def main():
    eng = cpi_amd.Engine(device=0)
    tag = os.path.basename(os.environ.get("CPI_AMD_LIB", "default"))
    for spec in sys.argv[1:]:
        name, W, lanes = spec.split(":")[:3]
        W, lanes = int(W), int(lanes)
        steps = int(spec.split(":")[3]) if spec.count(":") >= 3 else (200 if W <= 20000 else 10)
        kw = {}
        if os.environ.get("CPI_MB_POOL_BYTES"):      # e.g. 1 = a single batch, re-read from L2 / Infinity Cache
            kw["pool_bytes"] = int(os.environ["CPI_MB_POOL_BYTES"])
        N = int(os.environ.get("CPI_MB_SAMPLES", "50"))
        wl = bench.Workload(eng, name, W, N, seed=1234, lanes=lanes, **kw)
        best = 1e30
        for rep in range(3):
            k_ms = bench.time_steps(wl, steps, 5, graph=not os.environ.get("CPI_MB_EAGER"))["kernel_ms"]
            best = min(best, k_ms * 1e3 / steps)
        gbs = bench.bytes_per_unit(name, N) * W / (best * 1e-6) / 1e9
        print("%-18s %-10s W=%-8d L=%-3d launch_us=%10.2f  units/s=%.4g  GB/s=%.1f  frac=%.3f" % (
            tag, name, W, lanes, best, W / (best * 1e-6), gbs, gbs / 8000.0), flush=True)
        if wl.assembly:
            print("%-18s   assembly (cpi_assemble_tiles) %.4f ms per batch = %.0f GB/s of read + write" % (tag, wl.assembly["ms_per_batch"], wl.assembly["GBs"]), flush=True)
        del wl
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
