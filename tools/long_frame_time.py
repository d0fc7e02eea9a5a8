"""Stage times of the decoder on a few long frames (run on the GPU box): python tools/long_frame_time.py [kind] [MiB per frame] [frames] [level]"""
import ctypes, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.zeros(1, device="cuda")
import zstdsharp_amd as z, oracle_lib as o, datagen
from concurrent.futures import ThreadPoolExecutor
lib = z._ffi.load()
kind = sys.argv[1] if len(sys.argv) > 1 else "mixed"
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nfr = int(sys.argv[3]) if len(sys.argv) > 3 else 1
level = int(sys.argv[4]) if len(sys.argv) > 4 else 5
base = np.frombuffer(datagen.gen(kind, min(64 << 20, mib << 20), 7), dtype=np.uint8)
one = np.tile(base, ((mib << 20) + len(base) - 1) // len(base))[:mib << 20].tobytes()
t0 = time.time()
blob1 = o.compress(one, level, 0, 0)
print(f"oracle level {level}: {mib} MiB {kind} -> {len(blob1)} B in {time.time() - t0:.1f} s", flush=True)
blob = blob1 * nfr; n = (mib << 20) * nfr
comp = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).cuda()
want = torch.from_numpy(np.frombuffer(one, dtype=np.uint8).copy()).cuda().repeat(nfr)
out = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for mode in (0, 1) if mib * nfr <= 64 else (0,):
    d = z.Decompressor()
    lib.ZSTDMI_DCtx_setLongFrames(d.dctx, mode); lib.ZSTDMI_DCtx_setProfiling(d.dctx, 1)
    for rep in range(3):
        out.zero_(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = lib.ZSTDMI_decompressDevice(d.dctx, out.data_ptr(), n, comp.data_ptr(), comp.numel())
        dt = time.perf_counter() - t0
        assert r == n, lib.ZSTD_getErrorName(r)
    ms = (ctypes.c_float * 24)(); names = (ctypes.c_char_p * 24)()
    k = lib.ZSTDMI_DCtx_getStageTimes(d.dctx, ms, names, 24)
    st = {names[i].decode(): round(float(ms[i]), 3) for i in range(k)}
    print(f"mode {mode}: {dt * 1e3:.2f} ms wall = {n / dt / 1e9:.2f} GB/s, stages sum {sum(st.values()):.2f} ms, ok {bool(torch.equal(out, want))}\n   {st}", flush=True)
    d.Dispose()
