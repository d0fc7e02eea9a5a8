#!/usr/bin/env python3
"""Per-call latency of ZSTD_compress2 / ZSTD_decompressDCtx on small HOST buffers (the typical Wrap/Unwrap use)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, datagen
import zstdsharp_amd as z
lib = z._ffi.load()
c, d = z.Compressor(1), z.Decompressor()
for n in (4096, 65536, 1 << 20, 16 << 20):
    src = np.frombuffer(datagen.gen("text", n, 3), dtype=np.uint8).copy()
    cap = lib.ZSTD_compressBound(n); dst = np.empty(cap, dtype=np.uint8); back = np.empty(n, dtype=np.uint8)
    cs = lib.ZSTD_compress2(c.cctx, dst.ctypes.data, cap, src.ctypes.data, n)
    lib.ZSTD_decompressDCtx(d.dctx, back.ctypes.data, n, dst.ctypes.data, cs)
    reps = 50 if n <= (1 << 20) else 10
    t0 = time.perf_counter()
    for _ in range(reps): cs = lib.ZSTD_compress2(c.cctx, dst.ctypes.data, cap, src.ctypes.data, n)
    t1 = time.perf_counter()
    for _ in range(reps): r = lib.ZSTD_decompressDCtx(d.dctx, back.ctypes.data, n, dst.ctypes.data, cs)
    t2 = time.perf_counter()
    assert r == n and np.array_equal(back, src)
    print(f"{n:9d} B: compress {(t1 - t0) / reps * 1e6:8.1f} us ({n / ((t1 - t0) / reps) / 1e6:8.1f} MB/s)   decompress {(t2 - t1) / reps * 1e6:8.1f} us ({n / ((t2 - t1) / reps) / 1e6:8.1f} MB/s)", flush=True)
