"""Chunk-level data parallelism across the GPUs of one node.

Chunks are independent frames, so the path shards with no data-path collective: rank g of G owns the contiguous
chunk range [g*K/G, (g+1)*K/G) and the concatenation of the per-rank outputs in rank order IS the final stream.
The one exchange step is on the OUTPUT: an all-gather-v of the variable-length compressed shards
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist

CHUNK = 65536


def frame_span(level: int) -> int:
    """Input bytes per frame the library writes at `level` with its default framing: independent 64 KiB frames at the fast strategy
    (levels <= 2), five 48 KiB blocks at the doubleFast levels (3-4), eight 32 KiB blocks from level 5 on.  Shards cut on multiples
    of it so that the ranks' outputs, one behind the other, are the stream one GPU writes.  (Calls of 4 MiB or more at levels >= 3
    are first looked at in groups of 16 frames — the sparse-input probe —, so shards there are cut on 16 spans.)"""
    return CHUNK if level < 3 else 5 * (48 << 10) if level < 5 else 256 << 10


def shard_range(total_bytes: int, rank: int, world: int, chunk: int = CHUNK):
    """Byte range [lo, hi) of `total_bytes` owned by `rank`: whole units of `chunk` bytes (a frame's span, see frame_span),
    contiguous, in rank order."""
    k = (total_bytes + chunk - 1) // chunk
    lo_c, hi_c = rank * k // world, (rank + 1) * k // world
    return min(lo_c * chunk, total_bytes), min(hi_c * chunk, total_bytes)


def all_gather_sizes(nbytes: int, device, group=None) -> list:
    """Every rank's compressed shard size (one int64 per rank).  `group`: a separate process group keeps this tiny
    exchange from queueing behind a payload all-gather that is still in flight on the default group."""
    world = dist.get_world_size()
    mine = torch.tensor([nbytes], dtype=torch.int64, device=device)
    out = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return [int(v) for v in out.tolist()]


def all_gather_v(shard: torch.Tensor, nbytes: int, sizes: list, out: torch.Tensor = None, pad_to: int = None,
                 staging: torch.Tensor = None, method: str = "p2p", group=None):
    """All-gather variable-length byte shards.  Each rank contributes shard[:nbytes]; the result is the rank-order
    concatenation (a uint8 tensor of sum(sizes) bytes).

    method "p2p" (default): exact sizes, no padding, no compaction — every rank posts one send per peer and one receive
    per peer straight into that peer's offset of `out`, all in ONE group (ncclGroupStart/End under RCCL).  xGMI is a
    point-to-point mesh (7 links per GPU): G - 1 concurrent pair transfers use every link at once, where a ring
    all-gather moves all bytes over one link after the other.
    method "padded": ONE all_gather_into_tensor of slots padded to `pad_to` (or max(sizes)) + one batched copy that
    closes the gaps; kept for backends without grouped point-to-point."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    total = sum(sizes)
    if out is None or out.numel() < total:
        out = torch.empty(total, dtype=torch.uint8, device=shard.device)
    offs = [0] * (world + 1)
    for r, s in enumerate(sizes):
        offs[r + 1] = offs[r] + s
    assert sizes[rank] == nbytes, "sizes[] must hold what every rank contributes"
    if method == "p2p":
        ops = []
        for step in range(1, world):                     # peer order rotated per rank: no link is asked twice at once
            dst_r, src_r = (rank + step) % world, (rank - step) % world
            if nbytes:
                ops.append(dist.P2POp(dist.isend, shard[:nbytes], dst_r if group is None else dist.get_global_rank(group, dst_r), group))
            if sizes[src_r]:
                ops.append(dist.P2POp(dist.irecv, out[offs[src_r]:offs[src_r + 1]], src_r if group is None else dist.get_global_rank(group, src_r), group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        if nbytes:
            out[offs[rank]:offs[rank + 1]].copy_(shard[:nbytes])
        for q in reqs:
            q.wait()
        return out[:total]
    pad = pad_to if pad_to is not None else max(max(sizes), 1)
    assert shard.numel() >= pad, "the shard buffer must be at least as large as the padded slot"
    if staging is None or staging.numel() < world * pad:
        staging = torch.empty(world * pad, dtype=torch.uint8, device=shard.device)
    staging = staging[:world * pad]
    dist.all_gather_into_tensor(staging, shard[:pad].contiguous(), group=group)
    if total:
        torch.cat([staging[r * pad:r * pad + s] for r, s in enumerate(sizes) if s], out=out[:total])    # one batched copy
    return out[:total]
