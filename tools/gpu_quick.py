"""Quick GPU shake-down (development aid): round-trips through the HIP path, checked by the oracle decoder."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import zstdsharp_amd as z
import oracle_lib as o
import datagen

lib = z._ffi.load()
print("devices", lib.ZSTDMI_deviceCount(), flush=True)
c = z.Compressor(1); d = z.Decompressor()
bad = 0; tot = 0
sizes = [0, 1, 2, 6, 7, 8, 9, 63, 64, 100, 255, 256, 257, 1000, 1024, 4096, 16384, 65535, 65536, 65537, 131072, 200000, 1500000]
if len(sys.argv) > 1: sizes = [int(a) for a in sys.argv[1:]]
for kind in datagen.KINDS:
    for n in sizes:
        data = datagen.gen(kind, n, seed=n)
        tot += 1
        try:
            comp = c.Wrap(data)
        except Exception as e:
            print("COMPRESS FAIL", kind, n, e, flush=True); bad += 1; continue
        r = o.decompress(comp, len(data))
        ok1 = (r == data)
        try:
            r2 = d.Unwrap(comp); ok2 = (r2 == data)
        except Exception as e:
            ok2 = False; r2 = str(e)
        ref = o.compress(data, 1, 0, 65536)
        try:
            r3 = d.Unwrap(ref) if isinstance(ref, bytes) else None; ok3 = (r3 == data)
        except Exception as e:
            ok3 = False; r3 = str(e)
        if not (ok1 and ok2 and ok3):
            bad += 1
            print("FAIL", kind, n, "oracle-dec-of-gpu:", ok1 if ok1 else r if isinstance(r, int) else "mismatch",
                  "gpu-dec-of-gpu:", ok2 if ok2 else (r2 if isinstance(r2, str) else "mismatch"),
                  "gpu-dec-of-oracle:", ok3 if ok3 else (r3 if isinstance(r3, str) else "mismatch"), flush=True)
        elif n >= 65536:
            print(f"ok {kind:7s} {n:8d} gpu {len(comp):8d} oracle-chunked {len(ref):8d} ratio {len(comp)/max(n,1):.4f} vs {len(ref)/max(n,1):.4f}", flush=True)
print("total", tot, "bad", bad, flush=True)
