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

The default single-GPU run (no workload flags) also measures the other configurations BASELINE.json names, each at full size and
bit-exact, and files them under "extra_configs" of the same line (the headline keys are unchanged): synthetic text at level 1
(1 GiB, and 10 MiB = a dickens-sized call), the mixed corpus at level 5 (configs[2]), decompress-only of oracle-built level-5
frames (configs[4]: 4 GiB of 1 MiB frames, and one single 256 MiB frame).  Every entry carries value, stage_ms (one kernel per
stage), roofline and cpu_baseline; `--no-extra` skips them (tools/profile_round.sh profiles one workload per run).

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


# stage (one kernel each: the launchers mark the end of every kernel on the context's timer) -> kernel name prefixes in rocprofv3's tables
STAGE_KERNEL = {"compress/lz_fast": ("lz_kernel",), "compress/lz_region": ("lz_region_kernel",), "compress/huf_hist": ("huf_hist_kernel",),
                "compress/huf_tree": ("huf_tree_kernel",), "compress/huf_encode": ("huf_encode_kernel",),
                "compress/seq_encode": ("seq_encode_kernel",), "compress/gather": ("gather_kernel",), "compress/xxh64": ("xxh64_kernel",),
                "decompress/decode_literals": ("decode_literals_compact_kernel", "decode_literals_kernel", "decode_literals_sync_kernel"),
                "decompress/decode_literals_slow": ("decode_literals_slow_kernel",),
                "decompress/seq_decode": ("seq_decode_kernel",), "decompress/place_literals": ("place_literals_kernel",),
                "decompress/exec_matches": ("exec_matches_kernel",), "decompress/origin_init": ("origin_init_kernel",),
                "decompress/origin_jump": ("origin_jump_kernel",), "decompress/origin_gather": ("origin_gather_kernel",), "decompress/frame_walk": ("walk_segments_kernel",),
                "decompress/block_prepass": ("block_parse_kernel",), "decompress/block_offsets": ("block_offsets_kernel",)}


def pmc_traffic(stage, workload):
    """HBM bytes per launch of the stage's kernel from the newest committed rocprofv3 PMC summary TAKEN ON THIS WORKLOAD
    (tools/profile_round.sh: FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for gfx950; the summary records the bench line it was taken under, and config.workload must be the same string).  Else null."""
    import glob
    if stage not in STAGE_KERNEL:
        return None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
            if d["bench_line_under_profiler"]["config"]["workload"] != workload:
                continue
            best = None
            for name in STAGE_KERNEL[stage]:                       # (a stage whose kernel has several forms: the profile holds the one that ran)
                for k, v in d["kernels"].items():
                    if not k.startswith(name) or v.get("fetch_corrected") is None or v.get("write") is None:
                        continue
                    t = v["fetch_corrected"] + v["write"]
                    if best is None or t > best:
                        best = t
            if best is not None and best >= 1e5:
                return round(best / 1e9, 4), os.path.basename(path)
        except Exception:
            continue
    return None, None


def roofline(stages, alg_bytes, workload, n, t_comp_ms=None):
    """The slowest kernel of the step against the HBM roof: achieved = algorithmic bytes of one launch ((1 + r) bytes
    per uncompressed byte, whole buffer per launch) / that kernel's duration, measured with HIP events on the library's stream."""
    dom = max(stages, key=stages.get)
    achieved = alg_bytes / (stages[dom] * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(dom, workload)
    r = {"bound": "hbm", "kernel": dom, "kernel_ms": round(stages[dom], 4), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_unit": "GB per launch (PMC)", "traffic_source": traffic_src,
         "algorithmic_GB_per_launch": round(alg_bytes / 1e9, 4)}
    if t_comp_ms:       # north_star's wording: input bytes read per second by the whole COMPRESS side over the HBM peak
        r["hbm_read_frac_compress"] = round(n / (t_comp_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)
    return r


def stage_times(lib, ctx, getter):
    ms = (ctypes.c_float * 16)(); names = (ctypes.c_char_p * 16)()
    n = getter(ctx, ms, names, 16)
    return {names[i].decode(): float(ms[i]) for i in range(n)}


def oracle_frames(data: bytes, level: int, frame_bytes: int, threads: int):
    """`data` as independent zstd frames of frame_bytes each, built by the oracle (= the reference's algorithm at that level:
    128 KiB blocks, history and repeat-mode tables across the blocks of a frame) on `threads` host threads (ctypes drops the GIL)."""
    import oracle_lib as o
    from concurrent.futures import ThreadPoolExecutor
    pieces = [data[i:i + frame_bytes] for i in range(0, len(data), frame_bytes)]
    with ThreadPoolExecutor(max_workers=threads) as ex:
        out = list(ex.map(lambda b: o.compress(b, level, 0, 0), pieces))
    assert all(not isinstance(b, int) for b in out), "the oracle refused a frame"
    return b"".join(out), [len(b) for b in out]


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


class Env:
    """What every measurement needs: the library, torch, the device and (N > 1) the process group."""
    def __init__(self, lib, z, torch, dist, dev, rank, world, local):
        self.lib, self.z, self.torch, self.dist, self.dev, self.rank, self.world, self.local = lib, z, torch, dist, dev, rank, world, local
        self._base = {}

    def make_input(self, kind, nbytes, seed):
        import datagen, numpy as np
        torch = self.torch
        if kind == "zipf":
            return make_zipf(nbytes, seed, self.dev), "Zipf(alpha=1.1) bytes"
        key = (kind, min(64 << 20, nbytes), seed)
        if key not in self._base:
            if kind == "text":
                self._base[key] = datagen.text_like(key[1], seed)
            else:
                self._base[key] = np.frombuffer(datagen.gen("mixed", key[1], seed), dtype=np.uint8)
        base = self._base[key]
        what = ("synthetic text (declared stand-in for Silesia dickens, absent offline)" if kind == "text" else
                "synthetic mixed corpus: text, Zipf bytes, runs, random, periodic (declared stand-in for the concatenated Silesia corpus, absent offline)")
        t = torch.from_numpy(base).to(self.dev)
        reps = (nbytes + len(base) - 1) // len(base)
        return (t.repeat(reps)[:nbytes].contiguous() if reps > 1 else t[:nbytes].clone()), what

    def sync(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier(); self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_ok(self, ok):
        if self.dist is None:
            return ok
        t = self.torch.tensor([1 if ok else 0], device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item())


def run_decompress(env, args):
    """BASELINE configs[4]: decompress-only of pre-built frames.  The frames are built ONCE, outside the timed region: by the
    oracle (reference-shaped: one frame per --frame-mib of input, 128 KiB blocks chained by history, repcodes and repeat-mode
    tables) or by the GPU compressor (64 KiB single-block frames).  A step = one ZSTDMI_decompressDevice call over all of them,
    compressed input and output resident in HBM; value = regenerated bytes of all ranks / step time.  -> the JSON line (rank 0)"""
    import numpy as np
    lib, z, torch, dev = env.lib, env.z, env.torch, env.dev
    n = args.size_mib << 20
    unique = min(args.unique_mib << 20, n)
    frame_bytes = int(args.frame_mib * (1 << 20))
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    usrc, what = env.make_input(args.input, unique, 7 + env.rank)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.frames == "oracle":
        key = (args.input, unique, args.level, frame_bytes)           # (configurations that differ in total size share their frames)
        cache = env.__dict__.setdefault("_oracle_frames", {})
        if key not in cache:
            cache.clear()
            cache[key] = oracle_frames(usrc.cpu().numpy().tobytes(), args.level, frame_bytes, threads)
        blob, frame_sizes = cache[key]
        comp_u = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(dev)
        framing = f"oracle-built level-{args.level} frames of {args.frame_mib:g} MiB ({(frame_bytes + 131071) // 131072} blocks each)"
    else:
        blob = None
        cap = lib.ZSTD_compressBound(unique)
        tmp = torch.empty(cap, dtype=torch.uint8, device=dev)
        with z.Compressor(args.level, device=env.local) as c:
            cs = lib.ZSTDMI_compressDevice(c.cctx, tmp.data_ptr(), cap, usrc.data_ptr(), unique)
        assert cs < (1 << 63), lib.ZSTD_getErrorName(cs)
        comp_u = tmp[:cs].clone(); del tmp
        framing = f"GPU-built level-{args.level} frames, one per 64 KiB chunk"
    build_s = time.perf_counter() - t0
    reps = max(1, n // unique); n = reps * unique
    comp = comp_u.repeat(reps) if reps > 1 else comp_u
    want = usrc.repeat(reps) if reps > 1 else usrc
    csize = comp.numel()
    back = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    d = z.Decompressor(device=env.local)
    lib.ZSTDMI_DCtx_setProfiling(d.dctx, 1)

    def step():
        r = lib.ZSTDMI_decompressDevice(d.dctx, back.data_ptr(), n, comp.data_ptr(), csize)
        assert r == n, lib.ZSTD_getErrorName(r)

    for _ in range(args.warmup):
        step()
    env.sync()
    acc = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for k, v in stage_times(lib, d.dctx, lib.ZSTDMI_DCtx_getStageTimes).items(): acc[k] = acc.get(k, 0.0) + v
    env.sync()
    elapsed = env.max_over_ranks(time.perf_counter() - t0)
    ok = env.all_ok(bool(torch.equal(want, back)))
    d.Dispose()
    line = None
    if env.rank == 0:
        K = args.steps
        ratio = csize / n
        dec_ms = {"decompress/" + k: v / K for k, v in acc.items()}
        workload = f"decompress-only: {n >> 20} MiB of {what} per GPU as {framing}; {unique >> 20} MiB distinct, repeated x{reps}"
        line = {
            "metric": f"MB/s decompress, level-{args.level} frames", "value": round(env.world * n / (elapsed / K) / 1e6, 1) if ok else None, "unit": "MB/s",
            "n_gpus": env.world, "steps": K, "warmup": args.warmup, "ms_per_step": round(elapsed / K * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload,
                       "frames": n // frame_bytes if args.frames == "oracle" else (n + 65535) // 65536, "frame_build_s": round(build_s, 1),
                       "parallelism": f"replicas x{env.world} (no collective: frames are independent)"},
            "round_trip_bit_exact": ok, "ratio": round(ratio, 5),
            "decompress_MBps_per_gpu": round(n / (sum(dec_ms.values()) * 1e-3) / 1e6, 1),
            "stage_ms": {k: round(v, 4) for k, v in dec_ms.items()},
            "roofline": roofline(dec_ms, (1.0 + ratio) * n, workload, n),
        }
        if env.world == 1 and not args.no_cpu_baseline:
            import oracle_lib as o
            if args.frames == "oracle":         # whole frames of the same blob, up to the sample size (at least one)
                k = max(1, min(args.cpu_sample_mib << 20, unique) // frame_bytes)
                blob_s, m = blob[:sum(frame_sizes[:k])], min(k * frame_bytes, unique)
            else:
                blob_s, m = comp_u.cpu().numpy().tobytes(), unique
            t0 = time.perf_counter(); outb = o.decompress(blob_s, m); t1 = time.perf_counter()
            assert not isinstance(outb, int) and len(outb) == m
            line["cpu_baseline"] = {"value": round(m / (t1 - t0) / 1e6, 1), "unit": "MB/s", "cores": 1, "kind": "port",
                                    "sample": f"{m >> 20} MiB of the same frames decoded by oracle/ (C port of the reference's decoder)"}
    del back, comp, want, comp_u, usrc
    torch.cuda.empty_cache()
    return line, ok


def run_roundtrip(env, args):
    """The metric: one step = compress the whole buffer + decompress the result, both HBM-resident.  -> the JSON line (rank 0)"""
    lib, z, torch, dist, dev, rank, world, local = env.lib, env.z, env.torch, env.dist, env.dev, env.rank, env.world, env.local
    n = args.size_mib << 20
    src, what = env.make_input(args.input, n, (1234 if args.input == "zipf" else 7) + rank)
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
    out_buf = None
    step_no = 0

    def step(gather=True):
        nonlocal gathered, step_no, out_buf
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

    for _ in range(args.warmup):
        cs = step()
    env.sync()
    acc_c, acc_d = {}, {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cs = step()
        for k, v in stage_times(lib, c.cctx, lib.ZSTDMI_CCtx_getStageTimes).items(): acc_c[k] = acc_c.get(k, 0.0) + v
        for k, v in stage_times(lib, d.dctx, lib.ZSTDMI_DCtx_getStageTimes).items(): acc_d[k] = acc_d.get(k, 0.0) + v
    env.sync()
    elapsed = env.max_over_ranks(time.perf_counter() - t0)
    ok = bool(torch.equal(src, back))                  # correctness gate: no throughput without a bit-exact round trip
    if comm is not None and gathered is not None and rank == 0:
        last = dsts[(step_no - 1) % len(dsts)]
        ok = ok and bool(torch.equal(gathered[:cs], last[:cs]))
    ok = env.all_ok(ok)
    # N > 1: the collective-free number beside the gathered one (K more steps, outside the contract's timed region)
    elapsed_ng = None
    if comm is not None:
        env.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(gather=False)
        env.sync()
        elapsed_ng = env.max_over_ranks(time.perf_counter() - t0)

    line = None
    if rank == 0:
        K = args.steps
        ms_step = elapsed / K * 1e3
        ratio = cs / n
        comp_ms = {k: v / K for k, v in acc_c.items()}; dec_ms = {k: v / K for k, v in acc_d.items()}
        t_comp, t_dec = sum(comp_ms.values()), sum(dec_ms.values())
        allk = {**{"compress/" + k: v for k, v in comp_ms.items()}, **{"decompress/" + k: v for k, v in dec_ms.items()}}
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
            "roofline": roofline(allk, (1.0 + ratio) * n, workload, n, t_comp),      # (1 + r) bytes per input byte, both directions
        }
        if world == 1 and not args.no_cpu_baseline:
            m = min(args.cpu_sample_mib << 20, n)
            sample = src[:m].cpu().numpy().tobytes()
            threads = max(1, min(16, len(os.sched_getaffinity(0)))) if args.cpu_all_cores else 1
            one, allc = cpu_baseline(sample, threads, args.level)
            line["cpu_baseline"] = {"value": round(one["roundtrip"], 1), "unit": "MB/s", "cores": 1, "kind": "port",
                                    "sample": f"first {m >> 20} MiB of the same buffer, oracle/ (C port of the reference's level-{args.level} path), 64 KiB frames",
                                    "compress_MBps": round(one["compress"], 1), "decompress_MBps": round(one["decompress"], 1), "ratio": round(one["ratio"], 5),
                                    "all_cores": ({"value": round(allc["roundtrip"], 1), "cores": allc["cores"]} if allc else None)}
    c.Dispose(); d.Dispose()
    del src, back, dst, dsts
    torch.cuda.empty_cache()
    return line, ok


# What the default run measures besides the headline (name, overrides of the command-line defaults).  Steps are few: a step is a
# whole pass over the buffer (tens of milliseconds), and the whole default run has to stay within a couple of minutes.
EXTRA_CONFIGS = [
    ("text_l1_1gib", dict(input="text", steps=5, warmup=2, cpu_sample_mib=64)),
    ("text_l1_10mib_dickens_sized", dict(input="text", size_mib=10, steps=20, warmup=3, cpu_sample_mib=10)),
    ("mixed_l5_1gib_configs2", dict(input="mixed", level=5, steps=3, warmup=1, cpu_sample_mib=32)),
    ("decompress_l5_1mib_frames_1gib", dict(mode="decompress", input="mixed", level=5, frame_mib=1.0, size_mib=1024, unique_mib=256, steps=3, warmup=1, cpu_sample_mib=64)),
    ("decompress_l5_1mib_frames_4gib_configs4", dict(mode="decompress", input="mixed", level=5, frame_mib=1.0, size_mib=4096, unique_mib=256, steps=3, warmup=1, cpu_sample_mib=64)),
    ("decompress_l5_one_256mib_frame", dict(mode="decompress", input="mixed", level=5, frame_mib=256.0, size_mib=256, unique_mib=256, steps=2, warmup=1, cpu_sample_mib=256)),
]


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
    ap.add_argument("--no-extra", action="store_true", help="only the headline workload (the default single-GPU run also measures EXTRA_CONFIGS)")
    ap.add_argument("--parser", type=int, default=0, help="0 = region parse of dense chunks (default), 1 = tile loop only (ZSTDMI_CCtx_setParser)")
    ap.add_argument("--history", type=int, default=None, help="cross-chunk history in KiB per block (0 = independent 64 KiB frames; default: by level, "
                    "i.e. off at levels 1-2, 32 at levels >= 3)")
    args = ap.parse_args()
    args.cpu_all_cores = True
    defaults = ap.parse_args([])
    workload_flags = ("size_mib", "input", "mode", "frames", "frame_mib", "unique_mib", "level", "parser", "history")
    is_default_workload = all(getattr(args, k) == getattr(defaults, k) for k in workload_flags)

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
    env = Env(lib, z, torch, dist, dev, rank, world, local)

    line, ok = (run_decompress if args.mode == "decompress" else run_roundtrip)(env, args)
    if world == 1 and is_default_workload and not args.no_extra:
        import copy
        extras = []
        for name, over in EXTRA_CONFIGS:
            a = copy.copy(args)
            for k, v in over.items():
                setattr(a, k, v)
            a.cpu_all_cores = False
            t0 = time.perf_counter()
            try:
                e, eok = (run_decompress if a.mode == "decompress" else run_roundtrip)(env, a)
                e = {k: e[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "config", "round_trip_bit_exact", "ratio",
                                       "compress_MBps_per_gpu", "decompress_MBps_per_gpu", "stage_ms", "roofline", "cpu_baseline") if k in e}
            except Exception as ex:              # an extra configuration never takes the headline down with it
                e, eok = {"error": f"{type(ex).__name__}: {ex}"}, True
                torch.cuda.empty_cache()
            e["name"] = name; e["wall_s"] = round(time.perf_counter() - t0, 1)
            extras.append(e)
        line["extra_configs"] = extras
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()
    if not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
