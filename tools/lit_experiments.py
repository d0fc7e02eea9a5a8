"""Diagnostic: what the literal decoder waits for.  Timing-only builds (results are wrong by construction):
exp1 = no table lookup, exp2 = no stream loads in the loop, exp3 = (almost) no stores.  Build:
  for e in 1 2 3: hipcc ... -DZMI_LIT_EXPERIMENT=$e -shared -o zstdsharp_amd/libzstd_mi355x_exp$e.so csrc/*.hip"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.zeros(1, device="cuda")
import zstdsharp_amd._ffi as ffi
from bench import make_zipf, stage_times
base = ffi.load()
dev = torch.device("cuda", 0)
n = 1 << 30
src = make_zipf(n, 1234, dev); torch.cuda.synchronize()
cap = base.ZSTD_compressBound(n)
comp = torch.empty(cap + 8192, dtype=torch.uint8, device=dev); back = torch.empty(n, dtype=torch.uint8, device=dev)
c = base.ZSTD_createCCtx(); base.ZSTD_CCtx_setParameter(c, 100, 1)
cs = base.ZSTDMI_compressDevice(c, comp.data_ptr(), cap, src.data_ptr(), n)
for tag in ("", "_exp1", "_exp2", "_exp3"):
    path = os.path.join(ROOT, "zstdsharp_amd", f"libzstd_mi355x{tag}.so")
    if not os.path.exists(path): continue
    lib = ctypes.CDLL(path)
    for name, (res, args) in ffi.SIGNATURES.items():
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
    for mode in (3, 1):
        d = lib.ZSTD_createDCtx(); lib.ZSTDMI_DCtx_setProfiling(d, 1); lib.ZSTDMI_DCtx_setLiteralDecoder(d, mode)
        ts = []
        for _ in range(4):
            lib.ZSTDMI_decompressDevice(d, back.data_ptr(), n, comp.data_ptr(), cs)
            ts.append(stage_times(lib, d, lib.ZSTDMI_DCtx_getStageTimes)["decode_literals"])
        print(f"{tag or 'product':8s} decoder {mode}: decode_literals {min(ts):.3f} ms (runs {['%.3f' % t for t in ts]})", flush=True)
        lib.ZSTD_freeDCtx(d)
