"""Waves per frame in exec_matches (ZSTDMI_DCtx_setExecWaves): stage times on oracle-built frames.
python tools/exec_waves_time.py [kind] [frame MiB] [total MiB] [level]"""
import sys, os, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np, datagen
torch.zeros(1, device="cuda")
import zstdsharp_amd as z, oracle_lib as o
from concurrent.futures import ThreadPoolExecutor
if os.environ.get('ZMI_LIB'): z._ffi.LIB_PATH = os.path.join(ROOT, 'zstdsharp_amd', os.environ['ZMI_LIB'])     # a variant build (make variant)
lib = z._ffi.load()
kind = sys.argv[1] if len(sys.argv) > 1 else "mixed"; fm = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = int(sys.argv[3]) if len(sys.argv) > 3 else 1024; level = int(sys.argv[4]) if len(sys.argv) > 4 else 5
fb = int(fm * (1 << 20)); uniq = min(256, tot) << 20
base = np.frombuffer(datagen.gen(kind, min(64 << 20, uniq), 7), dtype=np.uint8)
data = np.tile(base, (uniq + len(base) - 1) // len(base))[:uniq].tobytes()
with ThreadPoolExecutor(16) as ex:
    parts = list(ex.map(lambda i: o.compress(data[i:i + fb], level, 0, 0), range(0, uniq, fb)))
reps = (tot << 20) // uniq
comp = torch.from_numpy(np.frombuffer(b"".join(parts), dtype=np.uint8).copy()).cuda().repeat(reps)
n = uniq * reps
out = torch.empty(n, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
for waves in (1, 2, 4, 8, 16, 0):
    d = lib.ZSTD_createDCtx(); lib.ZSTDMI_DCtx_setProfiling(d, 1); lib.ZSTDMI_DCtx_setLongFrames(d, 1); lib.ZSTDMI_DCtx_setExecWaves(d, waves)
    best = 1e9
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = lib.ZSTDMI_decompressDevice(d, out.data_ptr(), n, comp.data_ptr(), comp.numel()); assert r == n or os.environ.get('ZMI_LIB')
        best = min(best, time.perf_counter() - t0)
    ms = (ctypes.c_float * 24)(); names = (ctypes.c_char_p * 24)()
    k = lib.ZSTDMI_DCtx_getStageTimes(d, ms, names, 24)
    print(f"exec waves {waves}: best {best * 1e3:.2f} ms = {n / best / 1e9:.1f} GB/s", {names[i].decode(): round(ms[i], 3) for i in range(k)}, flush=True)
    lib.ZSTD_freeDCtx(d)
