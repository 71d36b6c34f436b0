"""Multi-GPU sharding of a batch of preintegration windows (one process per GPU).

Windows (and factors) are independent units: rank r owns the contiguous block
[r*ceil(W/n), (r+1)*ceil(W/n)) and runs the kernels on it with no data-path collective.  The only
exchange step is the FINAL gather of the per-rank output slabs TO THE ROOT (gather_to_root: ProcessGroupNCCL
issues one ncclSend per peer and the matching ncclRecvs on the root inside one group, i.e. every peer's slab
travels its own xGMI link; "gloo" in the CPU tests).  gather_packed / gather_outputs(dst=None) are the
all-gather forms for callers that need the results on every rank (n times the received bytes and memory).
Nothing like this exists in the reference (single-threaded CPU program); SURVEY.md section 8(e).  The
single-process C-ABI equivalent is cpi_group_gather (include/cpi_amd.h); chunk_bounds / assemble_chunks serve the exchange
INSIDE one batch (cpi_group_gather_chunk there).  Asking for the covariance as its packed upper triangle ("cov_sym": the field
P_sym, 120 instead of 225 doubles per window) shortens a full-V1 slab from 2 320 to 1 480 bytes per window.
"""
import torch
import torch.distributed as dist


def shard_bounds(W, rank, world):
    """Contiguous block partition with equal padded block size."""
    per = (W + world - 1) // world
    lo = min(W, rank * per)
    hi = min(W, lo + per)
    return lo, hi, per


def gather_outputs(local, W_total, group=None, dst=None):
    """local: dict name -> tensor [w_local, ...] (this rank's block, in shard_bounds order).
    Returns dict name -> tensor [W_total, ...] on every rank (dst=None, all_gather) or only on `dst`
    (others get None).  Slabs are padded to the common block size so one collective per field moves
    equal counts."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi, per = shard_bounds(W_total, rank, world)
    out = {}
    for name in sorted(local):
        t = local[name]
        assert t.shape[0] == hi - lo, (name, t.shape, lo, hi)
        pad = torch.zeros((per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: hi - lo] = t
        full = torch.empty((world * per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        if dst is None:
            dist.all_gather_into_tensor(full, pad, group=group)
            out[name] = full[:W_total]
        else:
            parts = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
            dist.gather(pad, parts, dst=dst, group=group)
            out[name] = torch.cat(parts, dim=0)[:W_total] if rank == dst else None
    return out


def pack_layout(fields, W):
    """Field-major layout of one rank's outputs inside ONE flat buffer: [name -> (offset, n)] in doubles, total size.
    fields = [(name, n)], every field [W, n] (n = 1 for [W])."""
    lay, off = {}, 0
    for name, n in fields:
        lay[name] = (off, n)
        off += n * W
    return lay, off


def alloc_packed(fields, W, device="cpu", dtype=torch.float64):
    """One flat buffer + per-field views of it ([W] for n = 1, else [W, n]): the layout Engine.alloc_outputs(packed=True)
    hands to the kernels, so that a rank's whole output is one contiguous slab.  Returns (flat, {name: view})."""
    lay, total = pack_layout(fields, W)
    flat = torch.empty((total,), dtype=dtype, device=device)
    views = {}
    for name, n in fields:
        v = flat[lay[name][0]:lay[name][0] + n * W]
        views[name] = v if n == 1 else v.view(W, n)
    return flat, views


def gather_packed(flat, fields, W_local, group=None):
    """The final gather as ONE collective: every rank's outputs live in one flat buffer (pack_layout; the engine can
    write into views of it directly, Engine.alloc_outputs(..., packed=True)), so the exchange step is a single
    all_gather_into_tensor of equal slabs instead of one per field -- with small batches the per-collective launch
    cost is what the gather costs.  Returns name -> tensor [world, W_local, n] (views of the gathered buffer; rank r's
    block is [r]).  Equal W_local on all ranks (weak scaling / padded blocks)."""
    world = dist.get_world_size(group)
    lay, total = pack_layout(fields, W_local)
    assert flat.numel() == total and flat.is_contiguous()
    full = torch.empty((world, total), dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(full.view(-1), flat, group=group)
    return {name: full[:, off:off + n * W_local].view(world, W_local, n) for name, (off, n) in lay.items()}


def gather_to_root(flat, fields, W_local, dst=0, group=None, out=None):
    """The final gather as SURVEY.md 8(e) asks for it: every rank's packed output slab (pack_layout; the engine writes
    into views of it, Engine.alloc_outputs(..., packed=True)) goes to rank `dst` only -- one collective, each peer
    sending straight to the root.  Returns name -> tensor [world, W_local, n] on the root (views of one [world, total]
    buffer; rank r's block is [r]) and None elsewhere.  `out`: optional preallocated [world, total] buffer on the root
    (a timed loop reuses it).  Equal W_local on all ranks (padded blocks)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lay, total = pack_layout(fields, W_local)
    assert flat.numel() == total and flat.is_contiguous()
    # gloo (the CPU test backend; also bench.py's single-device rehearsal) moves host memory: device slabs are staged
    staged = flat.is_cuda and dist.get_backend(group) == "gloo"
    src = flat.cpu() if staged else flat
    if rank == dst:
        full = out if out is not None else torch.empty((world, total), dtype=flat.dtype, device=flat.device)
        assert full.shape == (world, total) and full.is_contiguous()
        recv = torch.empty((world, total), dtype=flat.dtype) if staged else full
        dist.gather(src, [recv[r] for r in range(world)], dst=dst, group=group)
        if staged:
            full.copy_(recv)
        return {name: full[:, off:off + n * W_local].view(world, W_local, n) for name, (off, n) in lay.items()}
    dist.gather(src, None, dst=dst, group=group)
    return None


def chunk_bounds(W_local, chunk, chunks):
    """Sub-block `chunk` of `chunks` of a rank's block of W_local windows, relative to the block: (lo, hi, cper) with the equal
    sub-block size cper = ceil(W_local / chunks) -- the rule of cpi_shard_chunk_bounds (include/cpi_amd.h) for equal padded blocks.
    The exchange INSIDE one batch (bench.py --gather-schedule chunked; cpi_group_gather_chunk for single-process hosts): sub-block c's
    packed slab is gathered while sub-block c + 1 computes."""
    cper = -(-W_local // max(1, chunks))
    lo = min(W_local, chunk * cper)
    return lo, min(W_local, lo + cper), cper


def assemble_chunks(parts, W_local):
    """parts[c] = what gather_to_root returned for sub-block c (name -> [world, w_c, n]) -> name -> [world, W_local, n]."""
    return {name: torch.cat([p[name] for p in parts], dim=1)[:, :W_local] for name in parts[0]}


def unshard(gathered, W_total):
    """[world, W_local, n] blocks (gather_to_root / gather_packed) -> [W_total, n] in window order (padding dropped)."""
    return {name: t.reshape(-1, t.shape[-1])[:W_total] for name, t in gathered.items()}
