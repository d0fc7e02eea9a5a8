"""CPU test of the N>1 path: world_size-2 gloo processes exercise the sharding and the all-gather-v used by bench.py."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, total, q, cuts=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zstdsharp_amd.dist import shard_range, all_gather_sizes, all_gather_v
    data = (torch.arange(total, dtype=torch.int64) * 7 % 251).to(torch.uint8)
    # the shard every rank owns: the product's contiguous chunk ranges, or explicit uneven cuts (a zero-length shard included)
    ranges = [shard_range(total, r, world) for r in range(world)] if cuts is None else [(cuts[r], cuts[r + 1]) for r in range(world)]
    lo, hi = ranges[rank]
    # stand-in for "compress my shard": keep every third byte -> variable length per rank
    mine = data[lo:hi][::3].contiguous()
    buf = torch.zeros(hi - lo + 16, dtype=torch.uint8); buf[:mine.numel()] = mine
    sizes = all_gather_sizes(mine.numel(), "cpu")
    expect = torch.cat([data[a:b][::3] for a, b in ranges])
    ok = True
    for method in ("p2p", "padded"):
        pad = max(max(sizes), 1)
        if buf.numel() < pad:
            buf = torch.cat([buf, torch.zeros(pad - buf.numel(), dtype=torch.uint8)])
        out = all_gather_v(buf, mine.numel(), sizes, method=method)
        ok = ok and bool(torch.equal(out, expect))
    # the bench's form: sizes on their own process group, caller-owned (oversized) staging and output buffers, padded slots
    pg = dist.new_group(backend="gloo")
    sizes2 = all_gather_sizes(mine.numel(), "cpu", group=pg)
    pad2 = (max(sizes2) + 63) // 64 * 64
    if buf.numel() < pad2:
        buf = torch.cat([buf, torch.zeros(pad2 - buf.numel(), dtype=torch.uint8)])
    stage = torch.empty(world * pad2 + 1000, dtype=torch.uint8); outb = torch.empty(sum(sizes2) + 1000, dtype=torch.uint8)
    for method in ("p2p", "padded", "p2p"):  # buffers are reused step after step
        out2 = all_gather_v(buf, mine.numel(), sizes2, out=outb, pad_to=pad2, staging=stage, method=method)
        ok = ok and sizes2 == sizes and bool(torch.equal(out2, expect))
    q.put((rank, lo, hi, sizes, ok))
    dist.barrier(); dist.destroy_process_group()


def test_shard_ranges_partition_the_input():
    from zstdsharp_amd.dist import shard_range
    for total in (0, 1, 65536, 65537, 10 * 65536 + 5, (1 << 30)):
        for world in (1, 2, 3, 8):
            edges = [shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            for (a, b), (c, d) in zip(edges, edges[1:]):
                assert b == c and a <= b
            assert all(lo % 65536 == 0 for lo, _ in edges)
    from zstdsharp_amd.dist import frame_span
    for level, span in ((1, 65536), (3, 245760), (5, 262144)):
        assert frame_span(level) == span
        edges = [shard_range(10_000_000, r, 3, 16 * span) for r in range(3)]
        assert edges[0][0] == 0 and edges[-1][1] == 10_000_000 and all(lo % span == 0 for lo, _ in edges)


def test_all_gather_v_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    total = 5 * 65536 + 1234
    procs = [ctx.Process(target=_worker, args=(r, 2, 29641, total, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in procs)
    [p.join(60) for p in procs]
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == total
    assert res[0][3] == res[1][3]
    assert all(r[4] for r in res)


def test_all_gather_v_world4_uneven_and_empty_shards_gloo():
    """world 4, shards of very different sizes, one of them empty (a rank whose chunk range is empty when the input is short):
    both exchange forms must give the rank-order concatenation."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    total = 700001
    cuts = [0, 5, 5, 420000, total]                      # rank 1 owns nothing
    procs = [ctx.Process(target=_worker, args=(r, 4, 29653, total, q, cuts)) for r in range(4)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=180) for _ in procs)
    [p.join(60) for p in procs]
    assert [r[1:3] for r in res] == [(0, 5), (5, 5), (5, 420000), (420000, total)]
    assert res[1][3][1] == 0 and len(set(tuple(r[3]) for r in res)) == 1
    assert all(r[4] for r in res)
