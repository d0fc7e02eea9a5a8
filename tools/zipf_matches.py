#!/usr/bin/env python3
"""Diagnostic: what the match finder finds on the bench's Zipf data (sequences per 64 KiB chunk, their lengths and offsets)."""
import os, sys, ctypes
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import torch
torch.cuda.init()
import zstdsharp_amd as z
from zstdsharp_amd import _ffi
from bench import make_zipf
lib = _ffi.load()
n = 64 << 20
src = make_zipf(n, 1234, torch.device("cuda", 0)); torch.cuda.synchronize()
data = bytes(src.cpu().numpy())
c = z.Compressor(1)
comp = c.Wrap(data)
tot = 0; lens = {}; offs = []
for idx in range(0, 1024, 8):
    seqs = (_ffi.ZSTDMI_Seq * 16384)(); lits = ctypes.create_string_buffer(65536 + 512)
    ns, ls = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert lib.ZSTDMI_debugGetChunk(c.cctx, idx, seqs, 16384, ctypes.byref(ns), lits, 65536 + 512, ctypes.byref(ls)) == 0
    tot += ns.value
    for i in range(ns.value):
        ml = seqs[i].mlBase + 3; lens[ml] = lens.get(ml, 0) + 1; offs.append(seqs[i].offBase - 3)
print("sequences per chunk: %.2f over %d chunks" % (tot / 128, 128))
print("match lengths:", sorted(lens.items())[:12])
print("offsets: min %d median %d max %d" % (min(offs), sorted(offs)[len(offs) // 2], max(offs)) if offs else "none")
