"""Development aid: cross-chunk history mode (row f-1) — round trips under the oracle decoder and the GPU decoder, ratios beside
the independent-chunk mode and the oracle's 256 KiB-frame and unchunked outputs."""
import sys, os, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import zstdsharp_amd as z
import oracle_lib as o
import datagen

lib = z._ffi.load()


def pysrc(n):
    out = b""
    for f in sorted(glob.glob("/usr/lib/python3/dist-packages/**/*.py", recursive=True)):
        try: out += open(f, "rb").read()
        except OSError: pass
        if len(out) > n: break
    return out[:n]


bad = 0
d = z.Decompressor()
for kind, n in (("text", 3 << 20), ("text", (1 << 20) + 12345), ("mixed", 4 << 20), ("zipf", 1 << 20), ("runs", 1 << 20), ("period", 700001), ("pysrc", 4 << 20), ("text", 65537), ("text", 100000)):
    data = pysrc(n) if kind == "pysrc" else datagen.gen(kind, n, 5)
    n = len(data)
    for level in (1, 3, 5):
        row = []
        for hist, frame in ((0, 0), (16 << 10, 0), (32 << 10, 0), (32 << 10, 1 << 20), (48 << 10, 0)):
            c = z.Compressor(level)
            assert lib.ZSTDMI_CCtx_setHistory(c.cctx, hist, frame) == 0
            for chk in ((0, 1) if hist == (32 << 10) and frame == 0 else (0,)):
                c.SetParameter(201, chk)
                comp = c.Wrap(data)
                r1 = o.decompress(comp, n)
                ok1 = r1 == data
                try: ok2 = d.Unwrap(comp) == data
                except Exception as e: ok2 = str(e)
                if not (ok1 and ok2 is True):
                    bad += 1
                    print("FAIL", kind, n, "L", level, "hist", hist, "frame", frame, "chk", chk, "oracle:", ok1 if ok1 else (r1 if isinstance(r1, int) else "mismatch"), "gpu:", ok2, flush=True)
            row.append(round(len(comp) / n, 4))
        def osz(chunk):
            r = o.compress(data, level, 0, chunk)
            return len(r) / n if isinstance(r, bytes) else float("nan")
        ref256, ref0, ref64 = osz(262144), osz(0), osz(65536)
        print(f"{kind:7s} {n:8d} L{level} gpu hist 0/16K/32K/32K-1MiB-frames/48K: {row}  oracle 64K {ref64:.4f} 256K {ref256:.4f} unchunked {ref0:.4f}", flush=True)
print("bad", bad)
