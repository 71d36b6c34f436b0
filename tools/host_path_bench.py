"""PCIe-inclusive rate of the host-pointer entry (development tool; never the benchmarked path): pageable vs pinned host
memory, mean-only and "V1 full", per call of cpi_preintegrate_batch_host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cpi_amd
from cpi_amd import synth

eng = cpi_amd.Engine(device=0)
for W in (10000, 100000, 1000000):
    kn, lin, q = synth.make_windows(W, 50, seed=3, device=eng.device)
    kn, lin = kn.cpu(), lin.cpu()
    for pinned in (False, True):
        k2, l2 = (kn.pin_memory(), lin.pin_memory()) if pinned else (kn, lin)
        for want, nbytes in ((("mean",), 2944), (("mean", "jac", "cov"), 2856 + 2320)):
            prm = eng.make_params(1)
            out = eng.preintegrate_host(k2, l2, None, prm, want=want, pinned=pinned)   # the caller's buffers: allocated (and touched) once
            reps = 5 if W >= 1000000 else 20
            t0 = time.perf_counter()
            for _ in range(reps):
                eng.preintegrate_host(k2, l2, None, prm, want=want, pinned=pinned, out=out)
            dt = (time.perf_counter() - t0) / reps
            print("host entry  W=%-8d %-8s %-9s %8.3f ms per call  %6.1f M windows/s  %5.1f GB/s over PCIe (both directions summed)" % (
                W, "pinned" if pinned else "pageable", "full-V1" if len(want) > 1 else "mean", dt * 1e3, W / dt / 1e6, W * nbytes / dt / 1e9), flush=True)
