"""CPU-only, world_size 2, gloo: the N>1 path of the engine -- block partition of the windows and the
final gather of per-rank output slabs (cpi_amd/dist.py; RCCL all_gather over xGMI on the GPUs)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cpi_amd.dist import gather_outputs, gather_packed, pack_layout, shard_bounds  # noqa: F401


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


# --------------------------------------------------------------------------- bench.py's partition / gather code on engine-shaped outputs
def _worker_bench(rank, world, port, W, N, q):
    """What bench.py does at N > 1, with the oracle standing in for the kernels (no GPU here): block partition
    (cpi_amd.dist.shard_bounds), the rank's outputs written into the views of ONE packed slab (the layout
    Engine.alloc_outputs(packed=True) hands to the kernels), bench.final_gather to rank 0, max-over-ranks timing.
    Rank 0 checks the re-assembled result BITWISE against the unsharded run."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from cpi_amd import synth
        from cpi_amd.dist import alloc_packed, unshard
        from oracle import oracle_py as op
        kn, lin, qk = synth.make_windows(W, N, seed=606)
        kn, lin, qk = kn.numpy(), lin.numpy(), qk.numpy()
        ok = True
        for model, want in ((1, ("DT", "alpha", "beta", "q")), (2, ("DT", "alpha", "beta", "q", "J_q", "J_a", "J_b", "H_a", "H_b", "O_a", "O_b", "P"))):
            fields = [(n, k) for n, k in [("DT", 1), ("alpha", 3), ("beta", 3), ("q", 4), ("J_q", 9), ("J_a", 9), ("J_b", 9), ("H_a", 9),
                                          ("H_b", 9), ("O_a", 9), ("O_b", 9), ("P", 225)] if n in want]
            lo, hi, per = shard_bounds(W, rank, world)
            prm = op.make_params(model, 0, 1)
            loc = op.oracle().run(prm, kn[lo:hi], lin[lo:hi], qk[lo:hi])
            flat, views = alloc_packed(fields, per)              # padded block: equal slabs on every rank
            flat.zero_()
            for name, _ in fields:
                views[name][: hi - lo] = torch.from_numpy(np.ascontiguousarray(loc[name]))
            out = dict(views); out["_flat"], out["_fields"] = flat, fields
            for mode in ("root", "all"):
                g = bench.final_gather(out, per, mode, dst=0)
                if mode == "root" and rank != 0:
                    ok = ok and g is None
                    continue
                full = unshard(g, W)
                ref = op.oracle().run(prm, kn, lin, qk)
                for name, n in fields:
                    ok = ok and np.array_equal(full[name].numpy().reshape(W, -1), np.asarray(ref[name]).reshape(W, -1))
        t = torch.tensor([0.1 * (rank + 1), 7.0 - rank, 0.0], dtype=torch.float64)   # bench.main's (wall, kernel ms, wall w/o gather)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and abs(t[0].item() - 0.1 * world) < 1e-15 and t[1].item() == 7.0
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_bench_partition_and_gather_to_root_bitwise_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_bench, args=(r, 2, port, 77, 20, q)) for r in range(2)]   # 77: last block short
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _worker_chunked(rank, world, port, W, N, k, q):
    """The exchange INSIDE one batch (bench.py --gather-schedule chunked; cpi_group_gather_chunk in the C-ABI): every rank's block in k
    sub-blocks (cpi_amd.dist.chunk_bounds), each with a packed slab of its own that carries the covariance as its packed upper
    triangle (P_sym), gathered to rank 0 one sub-block at a time and put back together (assemble_chunks / unshard).  The oracle stands
    in for the kernels; rank 0 checks the result BITWISE against the unsharded run."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from cpi_amd import synth
        from cpi_amd.dist import alloc_packed, assemble_chunks, chunk_bounds, unshard
        from oracle import oracle_py as op
        kn, lin, qk = synth.make_windows(W, N, seed=707)
        kn, lin, qk = kn.numpy(), lin.numpy(), qk.numpy()
        cols = np.repeat(np.arange(15), np.arange(1, 16))
        rows = np.arange(120) - cols * (cols + 1) // 2
        pack = lambda P: np.ascontiguousarray(np.asarray(P).reshape(-1, 225)[:, cols * 15 + rows])   # CPI_TRI_INDEX
        fields = [("DT", 1), ("alpha", 3), ("beta", 3), ("q", 4), ("J_q", 9), ("H_a", 9), ("P_sym", 120)]
        lo, hi, per = shard_bounds(W, rank, world)
        prm = op.make_params(1, 0, 1)
        parts, ok, nmsg = [], True, 0
        for c in range(k):
            clo, chi, cper = chunk_bounds(per, c, k)
            a, b = min(hi, lo + clo), min(hi, lo + chi)          # a short last block: trailing sub-blocks may be short or empty
            flat, views = alloc_packed(fields, max(1, chi - clo))
            flat.zero_()
            if b > a:
                loc = op.oracle().run(prm, kn[a:b], lin[a:b], qk[a:b])
                loc["P_sym"] = pack(loc["P"])
                for name, _ in fields:
                    views[name][: b - a] = torch.from_numpy(np.ascontiguousarray(loc[name]).reshape(views[name][: b - a].shape))
            out = dict(views); out["_flat"], out["_fields"] = flat, fields
            g = bench.final_gather(out, max(1, chi - clo), "root", dst=0)
            nmsg += 1
            ok = ok and ((g is None) == (rank != 0))
            parts.append(g)
        if rank == 0:
            full = unshard(assemble_chunks(parts, per), W)
            ref = op.oracle().run(prm, kn, lin, qk)
            ref["P_sym"] = pack(ref["P"])
            for name, n in fields:
                ok = ok and np.array_equal(full[name].numpy().reshape(W, -1), np.asarray(ref[name]).reshape(W, -1))
        ok = ok and nmsg == k
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("W,k", [(77, 1), (77, 3), (80, 8), (5, 4)])
def test_chunked_gather_inside_one_batch_world2_gloo(W, k):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_chunked, args=(r, 2, port, W, 12, k, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_predicted_exchange_is_the_wire_budget_of_design_section_7():
    """bench.py's `predicted` object at N > 1: slab bytes per peer over ONE xGMI link at its one-way rate, and the expected wall time of
    the timed region under each schedule -- the numbers DESIGN.md section 7 states for configs[4] (1 M windows x full V1 per GPU)."""
    import bench

    class WL:
        W = 1000000
    for name, bytes_per_window in (("dense", 2248), ("sym", 1408), ("mean", 88)):
        wl = WL()
        fields = {"dense": [("DT", 1), ("alpha", 3), ("beta", 3), ("q", 4)] + [(n, 9) for n in ("J_q", "J_a", "J_b", "H_a", "H_b")] + [("P", 225)],
                  "sym": [("DT", 1), ("alpha", 3), ("beta", 3), ("q", 4)] + [(n, 9) for n in ("J_q", "J_a", "J_b", "H_a", "H_b")] + [("P_sym", 120)],
                  "mean": [("DT", 1), ("alpha", 3), ("beta", 3), ("q", 4)]}[name]
        wl.outs = [{"_fields": fields}]
        steps, k_ms = 3, 25.9
        for sched in ("final", "pipelined", "chunked"):
            p = bench.predicted_exchange(wl, 8, steps, sched, k_ms * steps, 8 * wl.W, chunks=8)
            assert p["slab_bytes_per_peer"] == wl.W * bytes_per_window and p["peers"] == 7 and p["link_GBs_one_way"] == 76.8
            e = wl.W * bytes_per_window / 76.8e9 * 1e3
            assert abs(p["exchange_ms_per_slab"] - e) < 1e-9 and abs(p["kernel_ms_per_step_measured"] - k_ms) < 1e-9
            want = {"final": steps * k_ms + e, "pipelined": k_ms + (steps - 1) * max(k_ms, e) + e,
                    "chunked": steps * max(k_ms, e) + min(k_ms, e) / 8}[sched]
            assert abs(p["expected_region_ms"] - want) < 1e-9
            assert abs(p["expected_value"] - 8 * wl.W * steps / (want * 1e-3)) < 1e-3
        # the statement of DESIGN 7 (model 1 writes 281 doubles per window, 176 with the packed covariance): dense P makes the 8-GPU
        # point root-ingress bound (29.3 ms of wire > 25.9 ms of kernels), packed P does not (18.3 ms)
        if name == "dense":
            assert 29.0 < e < 29.6 and p["exchange_bound"] is True
        if name == "sym":
            assert 18.0 < e < 18.6 and p["exchange_bound"] is False


def test_bench_gpus_flag_spawns_the_ranks_itself(monkeypatch):
    """`python bench.py --gpus N` with no launcher (WORLD_SIZE unset) re-executes under torch.distributed.run with N
    ranks on 127.0.0.1; under a launcher (WORLD_SIZE set) it does not."""
    import bench
    seen = {}

    def fake_execve(exe, cmd, env):
        seen["cmd"], seen["env"] = cmd, env
        raise SystemExit(0)
    monkeypatch.setattr(os, "execve", fake_execve)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr("sys.argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    try:
        bench.main()
    except SystemExit:
        pass
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
