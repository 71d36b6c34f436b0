"""Does replaying the launch loop from a hipGraph shorten the gap between consecutive launches of the headline batch?
(development measurement: plain stream launches vs one graph of `period` launches replayed)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import cpi_amd

W = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
eng = cpi_amd.Engine(device=0)
wl = bench.Workload(eng, "v1_mean", W, 50, seed=1)
n = len(wl.calls)
for _ in range(200):
    wl.step()
torch.cuda.synchronize()
steps = 4000
t0 = time.perf_counter()
for _ in range(steps):
    wl.step()
torch.cuda.synchronize()
plain = (time.perf_counter() - t0) / steps
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for c in wl.calls:
        c()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for c in wl.calls:
            c()
    g.replay(); torch.cuda.synchronize()
    reps = max(1, steps // n)
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / (reps * n)
print("W=%d  launches per graph %d   plain %.2f us/step   graph replay %.2f us/step" % (W, n, plain * 1e6, graph * 1e6))
