"""Development aid: replay the sequences + literals of every chunk of one input (ZSTDMI_debugGetChunk) the way the decoder
would and report the first inconsistency.  usage: dbg_replay.py <kind> <size> [seed]"""
import sys, os, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch; torch.cuda.init()
import datagen, zstdsharp_amd as z
from zstdsharp_amd import _ffi
import os
if os.environ.get('LIBZ'): _ffi.LIB_PATH = os.environ['LIBZ']
lib = z._ffi.load()
kind, n = sys.argv[1], int(sys.argv[2])
data = datagen.gen(kind, n, n + 3 if len(sys.argv) < 4 else int(sys.argv[3]))
c = z.Compressor(1)
comp = c.Wrap(data)
seqs = (_ffi.ZSTDMI_Seq * 16384)(); lits = ctypes.create_string_buffer(65536 + 512)
ns, ls = ctypes.c_size_t(0), ctypes.c_size_t(0)
for idx in range((n + 65535) // 65536):
    lo = idx * 65536; chunk = data[lo:lo + 65536]
    r = lib.ZSTDMI_debugGetChunk(c.cctx, idx, seqs, 16384, ctypes.byref(ns), lits, 65536 + 512, ctypes.byref(ls))
    L = lits.raw[:ls.value]
    out = bytearray(); rep = [1, 4, 8]; lp = 0; bad = None
    for i in range(ns.value):
        s = seqs[i]; ll, ml, ob = s.litLength, s.mlBase + 3, s.offBase
        out += L[lp:lp + ll]; lp += ll
        ll0 = 1 if ll == 0 else 0
        if ob > 3: off = ob - 3; rep = [off, rep[0], rep[1]]
        else:
            ix = ob - 1 + ll0
            if ix == 0: off = rep[0]
            else:
                off = rep[0] - 1 if ix == 3 else rep[ix]
                rep = [off, rep[0], rep[1]] if ix != 1 else [off, rep[0], rep[2]]
        if off <= 0 or off > len(out): bad = (i, "offset", off, len(out)); break
        for _ in range(ml): out.append(out[-off])
        if bytes(out) != chunk[:len(out)]:
            k = next(j for j in range(len(out)) if out[j] != chunk[j])
            bad = (i, "mismatch at", k, "seq pos", len(out) - ml, "ll", ll, "ml", ml, "off", off); break
    out += L[lp:]
    print("chunk", idx, "nbSeq", ns.value, "lits", ls.value, "replayed", len(out), "of", len(chunk), "BAD" if bad else "ok", bad)
    if bad:
        i = bad[0]
        pos = 0
        for j in range(max(0, i - 6), min(ns.value, i + 3)):
            s = seqs[j]; print("   seq", j, "ll", s.litLength, "ml", s.mlBase + 3, "offBase", s.offBase)
