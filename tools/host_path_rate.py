#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in boundary: ZSTD_compress2 / ZSTD_decompressDCtx on HOST buffers (what the C# caller
passes), staging through HBM inside the library.  Run on the GPU box."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, datagen
import zstdsharp_amd as z
lib = z._ffi.load()
n = 256 << 20
src = datagen.zipf_bytes(n, 3)
cap = lib.ZSTD_compressBound(n)
dst = np.empty(cap, dtype=np.uint8); back = np.empty(n, dtype=np.uint8)
c, d = z.Compressor(1), z.Decompressor()
for it in range(3):
    t0 = time.perf_counter()
    cs = lib.ZSTD_compress2(c.cctx, dst.ctypes.data, cap, src.ctypes.data, n)
    t1 = time.perf_counter()
    r = lib.ZSTD_decompressDCtx(d.dctx, back.ctypes.data, n, dst.ctypes.data, cs)
    t2 = time.perf_counter()
    assert r == n and np.array_equal(back, src)
    print(f"iter {it}: host compress {n / (t1 - t0) / 1e6:8.1f} MB/s   host decompress {n / (t2 - t1) / 1e6:8.1f} MB/s   round trip {n / (t2 - t0) / 1e6:8.1f} MB/s", flush=True)
