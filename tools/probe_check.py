"""Development aid: which inputs does the level >= 3 sparse probe send to level 1's finder?  (8 MiB per kind; equal sizes = level 1 taken)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import zstdsharp_amd as z, datagen
lib = z._ffi.load()
for kind in ("text", "mixed", "zipf", "rand", "runs", "period", "bytei"):
    data = datagen.gen(kind, 8 << 20, 3)
    row = {}
    for level, forced in ((1, False), (3, False), (5, False), (5, True)):
        with z.Compressor(level) as c:
            if forced: assert lib.ZSTDMI_CCtx_setHistory(c.cctx, 32 << 10, 0) == 0
            row[(level, forced)] = len(c.Wrap(data))
    print(f"{kind:7s} L1 {row[(1, False)]:9d}  L3 {row[(3, False)]:9d}  L5 {row[(5, False)]:9d}  L5 forced history {row[(5, True)]:9d}  -> level 5 took {'level 1' if row[(5, False)] == row[(1, False)] else 'its own'} path", flush=True)
