"""CPU-only, world_size 2, gloo: the N>1 path of the engine -- block partition of the windows and the
final gather of per-rank output slabs (cpi_amd/dist.py; RCCL all_gather over xGMI on the GPUs)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cpi_amd.dist import gather_outputs, gather_packed, pack_layout, shard_bounds


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
        # the one-collective flavour: the rank's outputs are views of one flat buffer
        Wl = 37
        fields = [("DT", 1), ("alpha", 3), ("q", 4)]
        lay, total = pack_layout(fields, Wl)
        flat = torch.zeros(total, dtype=torch.float64)
        gi = torch.arange(rank * Wl, (rank + 1) * Wl, dtype=torch.float64)
        flat[lay["DT"][0]:lay["DT"][0] + Wl] = gi
        flat[lay["alpha"][0]:lay["alpha"][0] + 3 * Wl].view(Wl, 3)[:] = gi[:, None] * torch.tensor([1.0, 2.0, 3.0])
        flat[lay["q"][0]:lay["q"][0] + 4 * Wl].view(Wl, 4)[:] = gi[:, None] + torch.arange(4.0)
        gp = gather_packed(flat, fields, Wl)
        gg = torch.arange(world * Wl, dtype=torch.float64)
        ok = ok and torch.equal(gp["DT"].reshape(-1), gg) and torch.equal(gp["alpha"].reshape(-1, 3)[:, 2], 3 * gg) \
            and torch.equal(gp["q"].reshape(-1, 4)[:, 3], gg + 3)
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
