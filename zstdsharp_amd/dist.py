"""Chunk-level data parallelism across the GPUs of one node.

Chunks are independent frames, so the path shards with no data-path collective: rank g of G owns the contiguous
chunk range [g*K/G, (g+1)*K/G) and the concatenation of the per-rank outputs in rank order IS the final stream.
The one exchange step is on the OUTPUT: an all-gather-v of the variable-length compressed shards
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist

CHUNK = 65536


def shard_range(total_bytes: int, rank: int, world: int, chunk: int = CHUNK):
    """Byte range [lo, hi) of `total_bytes` owned by `rank`: whole chunks, contiguous, in rank order."""
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
                 staging: torch.Tensor = None):
    """All-gather variable-length byte shards.  Each rank contributes shard[:nbytes]; the result is the rank-order
    concatenation (a uint8 tensor of sum(sizes) bytes).  Implemented as ONE padded all_gather_into_tensor (per-link
    bound on xGMI, so one large collective beats G small ones) followed by a local compaction."""
    world = dist.get_world_size()
    pad = pad_to if pad_to is not None else max(sizes)
    assert shard.numel() >= pad, "the shard buffer must be at least as large as the padded slot"
    if staging is None or staging.numel() < world * pad:
        staging = torch.empty(world * pad, dtype=torch.uint8, device=shard.device)
    staging = staging[:world * pad]
    dist.all_gather_into_tensor(staging, shard[:pad].contiguous())
    total = sum(sizes)
    if out is None or out.numel() < total:
        out = torch.empty(total, dtype=torch.uint8, device=shard.device)
    off = 0
    for r, s in enumerate(sizes):
        out[off:off + s] = staging[r * pad:r * pad + s]
        off += s
    return out[:total]
