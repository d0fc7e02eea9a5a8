#!/usr/bin/env python3
"""bench.py — the hot path on BASELINE.json's metric: MB/s compress+decompress at zstd level 1.

Workload (config.workload): BASELINE.json configs[1] — 1 GiB of i.i.d. Zipf(alpha=1.1) bytes per GPU, level 1,
64 KiB independent chunks, input already resident in HBM.  One "step" = compress the whole buffer with the HIP
pipeline and decompress the result with the HIP decoder (the metric names both directions); `value` is the
uncompressed bytes of all ranks divided by the step time.  Compress-only and decompress-only rates (HIP events
on the library's stream) are reported beside it.

N > 1 (driver launches one rank per GPU via torch.distributed.run): every rank owns its own shard of chunks
(weak scaling, no data-path collective on the input); the one real exchange step — the all-gather-v of the
compressed shards over RCCL/xGMI that BASELINE.json's north_star names — runs on a side stream, overlapped with
the decompress of its step and the compress of the next one (two output buffers alternate), and is inside the
timed region.  At N = 8 every rank receives 7 x 0.73 GiB per step, so the step time is the larger of the codec
time and the all-gather time.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def make_zipf(n, seed, device):
    import torch
    gen = torch.Generator(device=device); gen.manual_seed(seed)
    p = torch.arange(1, 257, dtype=torch.float64, device=device) ** -1.1
    cdf = torch.cumsum(p / p.sum(), 0).float()
    out = torch.empty(n, dtype=torch.uint8, device=device)
    step = 1 << 28
    for lo in range(0, n, step):
        m = min(step, n - lo)
        out[lo:lo + m] = torch.searchsorted(cdf, torch.rand(m, device=device, generator=gen)).clamp_(max=255).to(torch.uint8)
    return out


STAGE_KERNEL = {"compress/lz_fast": "lz_kernel", "compress/huf_build": "huf_hist_kernel", "compress/huf_encode": "huf_encode_kernel",
                "compress/seq_encode": "seq_encode_kernel", "compress/gather": "gather_kernel",
                "decompress/decode_literals": ("decode_literals_compact_kernel", "decode_literals_kernel", "decode_literals_sync_kernel"),
                "decompress/seq_decode": "seq_decode_kernel", "decompress/place_literals": "place_literals_kernel",
                "decompress/exec_matches": "exec_matches_kernel", "decompress/frame_walk": "walk_segments_kernel",
                "decompress/block_prepass": "block_parse_kernel", "decompress/block_offsets": "block_offsets_kernel"}


def pmc_traffic(stage, size_mib, kind, level):
    """HBM bytes per launch of the stage's kernel from the newest committed rocprofv3 PMC summary (tools/profile_round.sh:
    FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    Only valid for the workload the summary was taken on; otherwise null."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files or stage not in STAGE_KERNEL:
        return None, None
    try:
        d = json.load(open(files[-1]))
        w = d["bench_line_under_profiler"]["config"]["workload"]
        if not (w.startswith(f"{size_mib} MiB") and ("Zipf" in w) == (kind == "zipf") and f"level {level}" in w):
            return None, None
        names = STAGE_KERNEL[stage] if isinstance(STAGE_KERNEL[stage], tuple) else (STAGE_KERNEL[stage],)
        for name in names:                       # (the literal decoder has three forms; the profile holds the one that ran)
            for k, v in d["kernels"].items():
                if not k.startswith(name) or v.get("fetch_corrected") is None or v.get("write") is None:
                    continue
                if name == "lz_kernel" and not k.startswith("lz_kernel<" + ("0" if level <= 2 else "1" if level <= 4 else "2")):
                    continue
                if v["fetch_corrected"] + v["write"] < 1e6:
                    continue
                return round((v["fetch_corrected"] + v["write"]) / 1e9, 4), os.path.basename(files[-1])
    except Exception:
        pass
    return None, None


def stage_times(lib, ctx, getter):
    ms = (ctypes.c_float * 16)(); names = (ctypes.c_char_p * 16)()
    n = getter(ctx, ms, names, 16)
    return {names[i].decode(): float(ms[i]) for i in range(n)}


def oracle_frames(data: bytes, level: int, frame_bytes: int, threads: int) -> bytes:
    """`data` as independent zstd frames of frame_bytes each, built by the oracle (= the reference's algorithm at that level:
    128 KiB blocks, history and repeat-mode tables across the blocks of a frame) on `threads` host threads (ctypes drops the GIL)."""
    import oracle_lib as o
    from concurrent.futures import ThreadPoolExecutor
    pieces = [data[i:i + frame_bytes] for i in range(0, len(data), frame_bytes)]
    with ThreadPoolExecutor(max_workers=threads) as ex:
        out = list(ex.map(lambda b: o.compress(b, level, 0, 0), pieces))
    assert all(not isinstance(b, int) for b in out), "the oracle refused a frame"
    return b"".join(out)


def cpu_baseline(sample: bytes, threads: int, level: int = 1):
    """The oracle (a plain-C port of the reference's path at `level`), same 64 KiB framing, on the host cores."""
    import oracle_lib as o
    n = len(sample)
    t0 = time.perf_counter()
    comp = o.compress(sample, level, 0, 65536)
    t1 = time.perf_counter()
    back = o.decompress(comp, n)
    t2 = time.perf_counter()
    assert back == sample
    one = dict(compress=n / (t1 - t0) / 1e6, decompress=n / (t2 - t1) / 1e6, roundtrip=n / (t2 - t0) / 1e6, ratio=len(comp) / n)
    allc = None
    if threads > 1:
        per = (n // threads) // 65536 * 65536
        parts = [sample[i * per:(i + 1) * per] for i in range(threads)]
        def work(b):
            c = o.compress(b, level, 0, 65536); assert o.decompress(c, len(b)) == b
        ths = [threading.Thread(target=work, args=(b,)) for b in parts]
        t0 = time.perf_counter(); [t.start() for t in ths]; [t.join() for t in ths]; t1 = time.perf_counter()
        allc = dict(roundtrip=per * threads / (t1 - t0) / 1e6, cores=threads)
    return one, allc


def run_decompress(args, lib, z, torch, dist, dev, rank, world, local, make_input):
    """BASELINE configs[4]: decompress-only of pre-built frames.  The frames are built ONCE, outside the timed region: by the
    oracle (reference-shaped: one frame per --frame-mib of input, 128 KiB blocks chained by history, repcodes and repeat-mode
    tables) or by the GPU compressor (64 KiB single-block frames).  A step = one ZSTDMI_decompressDevice call over all of them,
    compressed input and output resident in HBM; value = regenerated bytes of all ranks / step time."""
    import numpy as np
    n = args.size_mib << 20
    unique = min(args.unique_mib << 20, n)
    frame_bytes = int(args.frame_mib * (1 << 20))
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    usrc, what = make_input(unique, 7 + rank)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.frames == "oracle":
        blob = oracle_frames(usrc.cpu().numpy().tobytes(), args.level, frame_bytes, threads)
        comp_u = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(dev)
        framing = f"oracle-built level-{args.level} frames of {args.frame_mib:g} MiB ({(frame_bytes + 131071) // 131072} blocks each)"
    else:
        cap = lib.ZSTD_compressBound(unique)
        tmp = torch.empty(cap, dtype=torch.uint8, device=dev)
        with z.Compressor(args.level, device=local) as c:
            cs = lib.ZSTDMI_compressDevice(c.cctx, tmp.data_ptr(), cap, usrc.data_ptr(), unique)
        assert cs < (1 << 63), lib.ZSTD_getErrorName(cs)
        comp_u = tmp[:cs].clone(); del tmp
        framing = f"GPU-built level-{args.level} frames, one per 64 KiB chunk"
    build_s = time.perf_counter() - t0
    reps = max(1, n // unique); n = reps * unique
    comp = comp_u.repeat(reps); want = usrc.repeat(reps)
    csize = comp.numel()
    back = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    d = z.Decompressor(device=local)
    lib.ZSTDMI_DCtx_setProfiling(d.dctx, 1)

    def step():
        r = lib.ZSTDMI_decompressDevice(d.dctx, back.data_ptr(), n, comp.data_ptr(), csize)
        assert r == n, lib.ZSTD_getErrorName(r)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(); torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    acc = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for k, v in stage_times(lib, d.dctx, lib.ZSTDMI_DCtx_getStageTimes).items(): acc[k] = acc.get(k, 0.0) + v
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); elapsed = float(t.item())
    ok = bool(torch.equal(want, back))
    if dist is not None:
        t = torch.tensor([1 if ok else 0], device=dev); dist.all_reduce(t, op=dist.ReduceOp.MIN); ok = bool(t.item())
    if rank == 0:
        K = args.steps
        ratio = csize / n
        dec_ms = {"decompress/" + k: v / K for k, v in acc.items()}
        dom = max(dec_ms, key=dec_ms.get)
        alg_bytes = (1.0 + ratio) * n
        achieved = alg_bytes / (dec_ms[dom] * 1e-3) / 1e9
        line = {
            "metric": f"MB/s decompress, level-{args.level} frames", "value": round(world * n / (elapsed / K) / 1e6, 1) if ok else None, "unit": "MB/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(elapsed / K * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"decompress-only: {n >> 20} MiB of {what} per GPU as {framing}; {unique >> 20} MiB distinct, repeated x{reps}",
                       "frames": n // frame_bytes if args.frames == "oracle" else (n + 65535) // 65536, "frame_build_s": round(build_s, 1),
                       "parallelism": f"replicas x{world} (no collective: frames are independent)"},
            "round_trip_bit_exact": ok, "ratio": round(ratio, 5),
            "decompress_MBps_per_gpu": round(n / (sum(dec_ms.values()) * 1e-3) / 1e6, 1),
            "stage_ms": {k: round(v, 4) for k, v in dec_ms.items()},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": None, "algorithmic_GB_per_launch": round(alg_bytes / 1e9, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            import oracle_lib as o
            m = min(args.cpu_sample_mib << 20, unique) // frame_bytes * frame_bytes if args.frames == "oracle" else min(args.cpu_sample_mib << 20, unique)
            blob_s = oracle_frames(usrc[:m].cpu().numpy().tobytes(), args.level, frame_bytes, threads) if args.frames == "oracle" \
                else comp_u.cpu().numpy().tobytes()
            m = m if args.frames == "oracle" else unique
            t0 = time.perf_counter(); outb = o.decompress(blob_s, m); t1 = time.perf_counter()
            assert not isinstance(outb, int) and len(outb) == m
            line["cpu_baseline"] = {"value": round(m / (t1 - t0) / 1e6, 1), "unit": "MB/s", "cores": 1, "kind": "port",
                                    "sample": f"{m >> 20} MiB of the same frames decoded by oracle/ (C port of the reference's decoder)"}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()
    d.Dispose()
    if not ok:
        sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size-mib", type=int, default=1024, help="uncompressed bytes per GPU (MiB)")
    ap.add_argument("--input", default="zipf", choices=["zipf", "text", "mixed"],
                    help="zipf: BASELINE configs[1]; text: stand-in for Silesia dickens; mixed: stand-in for the concatenated Silesia corpus (configs[2])")
    ap.add_argument("--mode", default="roundtrip", choices=["roundtrip", "decompress"],
                    help="roundtrip: the metric (compress + decompress per step); decompress: BASELINE configs[4], pre-built frames, decode only")
    ap.add_argument("--frames", default="oracle", choices=["oracle", "gpu"],
                    help="--mode decompress: who builds the frames (outside the timed region): the oracle = reference-shaped multi-block frames, or the GPU compressor")
    ap.add_argument("--frame-mib", type=float, default=1.0, help="--mode decompress --frames oracle: uncompressed bytes per frame (MiB)")
    ap.add_argument("--unique-mib", type=int, default=256, help="--mode decompress: distinct input the frames are built from (repeated up to --size-mib)")
    ap.add_argument("--level", type=int, default=1, help="compression level (BASELINE.json metric: 1; configs[3] uses 5)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the all-gather-v of compressed shards")
    ap.add_argument("--gather-method", default="p2p", choices=["p2p", "padded"], help="N>1: exact-size grouped send/recv, or one padded all-gather + compaction")
    ap.add_argument("--cpu-sample-mib", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parser", type=int, default=0, help="0 = region parse of dense chunks (default), 1 = tile loop only (ZSTDMI_CCtx_setParser)")
    ap.add_argument("--history", type=int, default=None, help="cross-chunk history in KiB per block (0 = independent 64 KiB frames; default: by level, "
                    "i.e. off at levels 1-2, 32 at levels >= 3)")
    args = ap.parse_args()

    import torch
    import zstdsharp_amd as z
    lib = z._ffi.load()
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print(f"bench.py --gpus {args.gpus} must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr); sys.exit(2)
    # torch touches the GPU first: its wheel bundles its own HIP runtime, which cannot enumerate the device once the system
    # runtime behind libzstd_mi355x.so holds it (the other order works; tests/conftest.py does the same)
    assert torch.cuda.is_available(), "no MI355X visible: the product has no CPU fallback"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.zeros(1, device=dev)
    assert lib.ZSTDMI_deviceCount() > local, "no MI355X visible to the library: the product has no CPU fallback"
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    n = args.size_mib << 20

    def make_input(nbytes, seed):
        import datagen, numpy as np
        if args.input == "zipf":
            return make_zipf(nbytes, seed, dev), "Zipf(alpha=1.1) bytes"
        if args.input == "text":
            base = datagen.text_like(min(64 << 20, nbytes), seed)
            what = "synthetic text (declared stand-in for Silesia dickens, absent offline)"
        else:
            base = np.frombuffer(datagen.gen("mixed", min(64 << 20, nbytes), seed), dtype=np.uint8)
            what = "synthetic mixed corpus: text, Zipf bytes, runs, random, periodic (declared stand-in for the concatenated Silesia corpus, absent offline)"
        return torch.from_numpy(np.tile(base, (nbytes + len(base) - 1) // len(base))[:nbytes].copy()).to(dev), what

    if args.mode == "decompress":
        run_decompress(args, lib, z, torch, dist, dev, rank, world, local, make_input)
        return
    src, what = make_input(n, (1234 if args.input == "zipf" else 7) + rank)
    hist_on = args.history > 0 if args.history is not None else args.level >= 3
    framing = ("256 KiB frames of " + ("64 KiB blocks, far matches up to 188 KiB back" if args.level < 3 else
                                       "48 KiB blocks behind 16 KiB of history (240 KiB frames)" if args.level < 5 and args.history is None else "32 KiB blocks behind 32 KiB of history")) if hist_on \
        else "one zstd frame per chunk"
    workload = f"{args.size_mib} MiB {what} per GPU, level {args.level}, " + ("cross-chunk history" if hist_on else "64 KiB independent chunks")
    torch.cuda.synchronize()          # the library runs on its own stream: the input must be complete before the first call
    cap = lib.ZSTD_compressBound(n)
    dst = torch.empty(cap + 8192, dtype=torch.uint8, device=dev)      # slack: the gather pads a shard up to a multiple of 4096
    back = torch.empty(n, dtype=torch.uint8, device=dev)
    c, d = z.Compressor(args.level, device=local), z.Decompressor(device=local)
    lib.ZSTDMI_CCtx_setProfiling(c.cctx, 1); lib.ZSTDMI_DCtx_setProfiling(d.dctx, 1)
    if args.history is not None:
        assert lib.ZSTDMI_CCtx_setHistory(c.cctx, args.history << 10, 0) == 0
    if args.parser:
        assert lib.ZSTDMI_CCtx_setParser(c.cctx, args.parser) == 0
    comm = torch.cuda.Stream(device=dev) if world > 1 and not args.no_gather else None
    gathered = None
    # N > 1: the all-gather-v of step i runs on a side stream and overlaps the decompress of step i and the compress of step
    # i + 1 (two output buffers alternate; a buffer is reused only after the gather that read it has finished).  Everything
    # stays inside the timed region; the sizes travel on their own process group so that they do not queue behind a payload.
    dsts = [dst, torch.empty_like(dst)] if comm is not None else [dst]
    gather_done = [None, None]
    pg_sizes = dist.new_group(backend="nccl") if comm is not None else None
    stage_buf = out_buf = None
    step_no = 0

    def step(gather=True):
        nonlocal gathered, step_no, stage_buf, out_buf
        b = step_no % len(dsts); step_no += 1
        buf = dsts[b]
        if gather_done[b] is not None:
            gather_done[b].synchronize()
        cs = lib.ZSTDMI_compressDevice(c.cctx, buf.data_ptr(), cap, src.data_ptr(), n)
        assert cs < (1 << 63), lib.ZSTD_getErrorName(cs)
        if comm is not None and gather:
            from zstdsharp_amd.dist import all_gather_sizes, all_gather_v
            sizes = all_gather_sizes(cs, dev, group=pg_sizes)
            comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(comm):
                if out_buf is None or out_buf.numel() < sum(sizes):
                    out_buf = torch.empty(sum(sizes) + (64 << 20), dtype=torch.uint8, device=dev)
                # exact-size grouped send/recv: every pair's bytes travel on that pair's own xGMI link, no padding, no compaction
                gathered = all_gather_v(buf, cs, sizes, out=out_buf, method=args.gather_method)
                ev = torch.cuda.Event(); ev.record(comm); gather_done[b] = ev
        r = lib.ZSTDMI_decompressDevice(d.dctx, back.data_ptr(), n, buf.data_ptr(), cs)
        assert r == n, lib.ZSTD_getErrorName(r)
        return cs

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        cs = step()
    sync()
    acc_c, acc_d = {}, {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cs = step()
        for k, v in stage_times(lib, c.cctx, lib.ZSTDMI_CCtx_getStageTimes).items(): acc_c[k] = acc_c.get(k, 0.0) + v
        for k, v in stage_times(lib, d.dctx, lib.ZSTDMI_DCtx_getStageTimes).items(): acc_d[k] = acc_d.get(k, 0.0) + v
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); elapsed = float(t.item())
    ok = bool(torch.equal(src, back))                  # correctness gate: no throughput without a bit-exact round trip
    if comm is not None and gathered is not None and rank == 0:
        last = dsts[(step_no - 1) % len(dsts)]
        ok = ok and bool(torch.equal(gathered[:cs], last[:cs]))
    if dist is not None:
        t = torch.tensor([1 if ok else 0], device=dev); dist.all_reduce(t, op=dist.ReduceOp.MIN); ok = bool(t.item())
    # N > 1: the collective-free number beside the gathered one (K more steps, outside the contract's timed region)
    elapsed_ng = None
    if comm is not None:
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(gather=False)
        sync()
        elapsed_ng = time.perf_counter() - t0
        t = torch.tensor([elapsed_ng], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); elapsed_ng = float(t.item())

    if rank == 0:
        K = args.steps
        ms_step = elapsed / K * 1e3
        ratio = cs / n
        comp_ms = {k: v / K for k, v in acc_c.items()}; dec_ms = {k: v / K for k, v in acc_d.items()}
        t_comp, t_dec = sum(comp_ms.values()), sum(dec_ms.values())
        allk = {**{"compress/" + k: v for k, v in comp_ms.items()}, **{"decompress/" + k: v for k, v in dec_ms.items()}}
        dom = max(allk, key=allk.get)
        alg_bytes = (1.0 + ratio) * n                  # (1 + r) bytes per input byte, both directions
        achieved = alg_bytes / (allk[dom] * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(dom, args.size_mib, args.input, args.level)
        line = {
            "metric": f"MB/s compress+decompress, level {args.level}", "value": round(world * n / (elapsed / K) / 1e6, 1) if ok else None, "unit": "MB/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "chunk": 65536, "framing": framing,
                       "gather": (f"all-gather-v of compressed shards over RCCL ({args.gather_method}) on a side stream, overlapped with the decompress of its step and the compress of the next" if comm is not None else "none"),
                       "parallelism": f"chunk-sharded x{world}"},
            "round_trip_bit_exact": ok, "ratio": round(ratio, 5),
            "value_no_gather": (round(world * n / (elapsed_ng / K) / 1e6, 1) if elapsed_ng and ok else None),
            "compress_MBps_per_gpu": round(n / (t_comp * 1e-3) / 1e6, 1), "decompress_MBps_per_gpu": round(n / (t_dec * 1e-3) / 1e6, 1),
            "stage_ms": {k: round(v, 4) for k, v in allk.items()},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_unit": "GB per launch (PMC)", "traffic_source": traffic_src,
                         "algorithmic_GB_per_launch": round(alg_bytes / 1e9, 4),
                         "hbm_read_frac": round(n / (allk[dom] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)},
        }
        if world == 1 and not args.no_cpu_baseline:
            m = min(args.cpu_sample_mib << 20, n)
            sample = src[:m].cpu().numpy().tobytes()
            threads = max(1, min(16, len(os.sched_getaffinity(0))))
            one, allc = cpu_baseline(sample, threads, args.level)
            line["cpu_baseline"] = {"value": round(one["roundtrip"], 1), "unit": "MB/s", "cores": 1, "kind": "port",
                                    "sample": f"first {m >> 20} MiB of the same buffer, oracle/ (C port of the reference's level-{args.level} path), 64 KiB frames",
                                    "compress_MBps": round(one["compress"], 1), "decompress_MBps": round(one["decompress"], 1), "ratio": round(one["ratio"], 5),
                                    "all_cores": ({"value": round(allc["roundtrip"], 1), "cores": allc["cores"]} if allc else None)}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()
    c.Dispose(); d.Dispose()
    if not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
