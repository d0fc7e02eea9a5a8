#!/usr/bin/env python3
"""GPU vs oracle compressed sizes per level on the synthetic kinds (run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen, oracle_lib as o
import zstdsharp_amd as z
if os.environ.get('ZMI_LIB'):
    z._ffi.LIB_PATH = os.path.join(ROOT, 'zstdsharp_amd', os.environ['ZMI_LIB'])

import glob
def corpus(kind):
    if kind == "pysrc":       # tuning aid only: the image's own python sources as a stand-in for real source/text files
        out = bytearray()
        for f in sorted(glob.glob("/usr/lib/python3.10/*.py")):
            out += open(f, "rb").read()
            if len(out) >= (4 << 20): break
        return bytes(out[:4 << 20])
    if kind == "licenses":
        out = bytearray()
        for f in sorted(glob.glob("/usr/share/common-licenses/*")) + sorted(glob.glob("/usr/share/doc/*/copyright")):
            try: out += open(f, "rb").read()
            except OSError: pass
            if len(out) >= (4 << 20): break
        return bytes(out[:4 << 20])
    return datagen.gen(kind, 1 << 20, 21)

for kind in ("pysrc", "licenses", "text", "mixed", "zipf", "runs"):
    data = corpus(kind)
    if not data: continue
    row = []
    for level in (1, 3, 4, 5, 7):
        with z.Compressor(level) as c:
            g = len(c.Wrap(data))
        r = len(o.compress(data, level, 0, 65536)) if level <= 5 else 0
        row.append(f"L{level}: gpu {g} ref {r}")
    print(kind, " | ".join(row), flush=True)
