import sys, os, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch; torch.cuda.init()
import datagen, zstdsharp_amd as z
from zstdsharp_amd import _ffi
lib = z._ffi.load()
data = datagen.gen("runs", 1 << 20, 6)
c = z.Compressor(1)
comp = c.Wrap(data)
print("size", len(comp))
seqs = (_ffi.ZSTDMI_Seq * 16384)(); lits = ctypes.create_string_buffer(65536 + 512)
ns, ls = ctypes.c_size_t(0), ctypes.c_size_t(0)
for idx in (0, 1):
    r = lib.ZSTDMI_debugGetChunk(c.cctx, idx, seqs, 16384, ctypes.byref(ns), lits, 65536 + 512, ctypes.byref(ls))
    print("chunk", idx, "nbSeq", ns.value, "lits", ls.value)
    pos = 0
    for i in range(min(ns.value, 60)):
        s = seqs[i]; pos += s.litLength
        print(f"  seq {i}: pos {pos} ll {s.litLength} ml {s.mlBase + 3} offBase {s.offBase}")
        pos += s.mlBase + 3
