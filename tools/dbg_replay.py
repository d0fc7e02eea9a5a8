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
    if len(out) != len(chunk):
        extra = len(out) - len(chunk)
        tail = L[-extra - 8:]
        print("   last literal bytes", tail.hex(), "| chunk tail", chunk[-16:].hex())
        # where do the final `extra` literal bytes come from?
        t = L[-extra:]
        idxs = [i for i in range(len(chunk) - len(t) + 1) if chunk[i:i + len(t)] == t][:5]
        print("   extra bytes occur in chunk at", idxs, "trailing literals expected", len(chunk) - (len(out) - len(L[lp:])), "got", len(L[lp:]))
        pos = 0
        for j in range(ns.value):
            s = seqs[j]; pos += s.litLength
            if j >= ns.value - 6: print("   seq", j, "start", pos, "region", (pos - 4096) // 64, "q", (pos - 4096) % 64, "ll", s.litLength, "ml", s.mlBase + 3, "offBase", s.offBase, "end", pos + s.mlBase + 3)
            pos += s.mlBase + 3

if os.environ.get("DBGKEEP"):
    import struct
    raw = ctypes.CDLL(_ffi.LIB_PATH); raw.ZSTDMI_debugReadCand.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t]
    nreg = (min(n, 4096 + 30720) - 4096 + 63) // 64
    buf = ctypes.create_string_buffer(nreg * 32)
    raw.ZSTDMI_debugReadCand(c.cctx, buf, 4096 * 2, nreg * 32)
    # expected keep per region from the sequences
    cov = bytearray(len(data)); pos = 0
    for j in range(ns.value):
        s = seqs[j]; pos += s.litLength
        for t in range(s.mlBase + 3): cov[pos + t] = 1
        pos += s.mlBase + 3
    for r in range(nreg):
        keep, selLo, selHi, ee = struct.unpack_from("<QQQQ", buf.raw, r * 32)
        rs_ = 4096 + 64 * r
        exp = 0
        for b in range(64):
            if rs_ + b < len(data) and not cov[rs_ + b]: exp |= 1 << b
        if exp != keep:
            print("region", r, "rs", rs_, "keep", hex(keep), "expected", hex(exp), "selLo", hex(selLo), "selHi", hex(selHi), "eSpec", ee >> 32, "lastEnd", ee & 0xFFFFFFFF)
    pos = 0
    for j in range(ns.value):
        s = seqs[j]; pos += s.litLength
        if 16100 <= pos <= 16500: print("   seq", j, "start", pos, "region", (pos - 4096) // 64, "q", (pos - 4096) % 64, "ll", s.litLength, "ml", s.mlBase + 3, "offBase", s.offBase, "end", pos + s.mlBase + 3)
        pos += s.mlBase + 3
    for r in range(186, 194):
        keep, selLo, selHi, ee = struct.unpack_from("<QQQQ", buf.raw, r * 32)
        print("   region", r, "rs", 4096 + 64 * r, "keep", hex(keep), "selLo", hex(selLo), "selHi", hex(selHi), "eSpec", ee >> 32, "lastEnd", ee & 0xFFFFFFFF)
