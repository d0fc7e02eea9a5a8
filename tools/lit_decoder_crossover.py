#!/usr/bin/env python3
"""Decompress time of the two literal decoders against input size (run on the GPU box)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen
import zstdsharp_amd as z
lib = z._ffi.load()
full = torch.from_numpy(datagen.zipf_bytes(1 << 30, 3).copy()).cuda(); torch.cuda.synchronize()
for mib in (1, 4, 16, 64, 128, 256, 384, 512, 1024):
    n = mib << 20; src = full[:n]
    cap = lib.ZSTD_compressBound(n); dst = torch.empty(cap, dtype=torch.uint8, device="cuda"); back = torch.empty(n, dtype=torch.uint8, device="cuda")
    c, d = z.Compressor(1), z.Decompressor()
    cs = lib.ZSTDMI_compressDevice(c.cctx, dst.data_ptr(), cap, src.data_ptr(), n)
    lib.ZSTDMI_DCtx_setProfiling(d.dctx, 1)
    row = []
    for mode in (1, 2, 3):
        lib.ZSTDMI_DCtx_setLiteralDecoder(d.dctx, mode)
        best = 1e9
        for _ in range(3):
            r = lib.ZSTDMI_decompressDevice(d.dctx, back.data_ptr(), n, dst.data_ptr(), cs); assert r == n
            ms = (ctypes.c_float * 16)(); names = (ctypes.c_char_p * 16)(); k = lib.ZSTDMI_DCtx_getStageTimes(d.dctx, ms, names, 16)
            t = {names[i].decode(): ms[i] for i in range(k)}
            best = min(best, t["decode_literals"])
        row.append(best)
    print(f"{mib:5d} MiB  frames {n >> 16:6d}  serial {row[0]:8.3f} ms   selfsync {row[1]:8.3f} ms   compact {row[2]:8.3f} ms", flush=True)
    c.Dispose(); d.Dispose()
