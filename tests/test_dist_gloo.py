"""CPU-only, world_size 2, gloo: the N>1 path of the engine -- block partition of the windows and the
final gather of per-rank output slabs (cpi_amd/dist.py; RCCL all_gather over xGMI on the GPUs)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cpi_amd.dist import gather_outputs, shard_bounds


def test_shard_bounds_cover_exactly():
    for W in (0, 1, 7, 8, 9, 10000, 100001):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                lo, hi, per = shard_bounds(W, r, world)
                assert 0 <= lo <= hi <= W and hi - lo <= per
                seen += list(range(lo, hi))
            assert seen == list(range(W))


def _worker(rank, world, port, W, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi, per = shard_bounds(W, rank, world)
        # a stand-in for the per-rank kernel outputs: a deterministic function of the GLOBAL window index
        idx = torch.arange(lo, hi, dtype=torch.float64)
        local = {"alpha": torch.stack([idx, 2 * idx, 3 * idx], dim=1), "DT": idx * 0.5,
                 "P": idx[:, None] * torch.ones((1, 225), dtype=torch.float64)}
        full = gather_outputs(local, W)                       # all_gather flavour
        g = torch.arange(W, dtype=torch.float64)
        ok = torch.equal(full["DT"], g * 0.5) and torch.equal(full["alpha"][:, 2], 3 * g) and torch.equal(full["P"][:, 7], g)
        root = gather_outputs(local, W, dst=0)                # gather-to-root flavour
        if rank == 0:
            ok = ok and torch.equal(root["alpha"][:, 1], 2 * g)
        else:
            ok = ok and root["alpha"] is None
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and t.item() == float(world)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_gather_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    W = 1001                                                   # odd: last rank's block is short (padding path)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, W, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
