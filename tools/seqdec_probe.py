"""Development aid: stage times of one decompress call on 256 MiB of text frames, whatever the call returns (used with
experimental builds of the library that decode garbage on purpose: LIB=path python tools/seqdec_probe.py)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np, datagen
import zstdsharp_amd._ffi as ffi
good = ffi.load()
n = 256 << 20
host = np.tile(datagen.text_like(32 << 20, 7), 8)[:n]
src = torch.from_numpy(host.copy()).cuda(); torch.cuda.synchronize()
cap = good.ZSTD_compressBound(n); dst = torch.empty(cap, dtype=torch.uint8, device="cuda"); back = torch.empty(n, dtype=torch.uint8, device="cuda")
c = good.ZSTD_createCCtx(); good.ZSTD_CCtx_setParameter(c, 100, int(os.environ.get("LEVEL", "1")))
cs = good.ZSTDMI_compressDevice(c, dst.data_ptr(), cap, src.data_ptr(), n)
for path in [ffi.LIB_PATH] + ([os.environ["LIB"]] if os.environ.get("LIB") else []):
    lib = ctypes.CDLL(path)
    for name, (res, args) in ffi.SIGNATURES.items():
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
    d = lib.ZSTD_createDCtx(); lib.ZSTDMI_DCtx_setProfiling(d, 1)
    for _ in range(3):
        r = lib.ZSTDMI_decompressDevice(d, back.data_ptr(), n, dst.data_ptr(), cs)
    ms = (ctypes.c_float * 32)(); names = (ctypes.c_char_p * 32)()
    k = lib.ZSTDMI_DCtx_getStageTimes(d, ms, names, 32)
    print(os.path.basename(path), "ok" if r == n else f"ret {r:#x}", {names[i].decode(): round(ms[i] * 4, 3) for i in range(k)}, "(ms per GiB)", flush=True)
