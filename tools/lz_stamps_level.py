"""Diagnostic: phase shares of lz_kernel at a given level / input from the stamped build (make -C zstdsharp_amd/csrc stamps).
usage: python tools/lz_stamps_level.py LEVEL [text|mixed|zipf ...]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen, numpy as np
import zstdsharp_amd._ffi as ffi
ffi.LIB_PATH = os.path.join(ROOT, "zstdsharp_amd", "libzstd_mi355x_stamps.so")
lib = ffi.load()
raw = ctypes.CDLL(ffi.LIB_PATH); raw.ZSTDMI_debugReadLzStamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
names = ["stage", "probe(pre-A)", "barrier A", "verify(phase B)", "barrier B", "emit(+sparse)", "barrier C", "literals", "history insert | dense tile: select", "dense tile: finish+rank", "region: candidates (I)", "region: load pass", "region: speculative parse", "region: real parse", "region: links+scans", "region: emit+literals", "hc: hashes", "hc: barrier", "hc: links (wave 0)", "hc: barrier", "hc: search", "hc: barrier", "hc: ring copy + barrier/store drain", "-"]
level = int(sys.argv[1]); kinds = sys.argv[2:] or ["text"]
n = 256 << 20
for kind in kinds:
    if kind == "zipf": host = datagen.zipf_bytes(n, 3)
    elif kind == "text": host = np.tile(datagen.text_like(32 << 20, 7), 8)[:n]
    else: host = np.frombuffer(datagen.gen(kind, 32 << 20, 5) * 8, dtype=np.uint8)[:n]
    src = torch.from_numpy(host.copy()).cuda(); torch.cuda.synchronize()
    cap = lib.ZSTD_compressBound(n); dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    c = lib.ZSTD_createCCtx(); lib.ZSTD_CCtx_setParameter(c, 100, level)
    lib.ZSTDMI_compressDevice(c, dst.data_ptr(), cap, src.data_ptr(), n)
    buf = (ctypes.c_ulonglong * 24)(); raw.ZSTDMI_debugReadLzStamps(buf, 1)
    lib.ZSTDMI_compressDevice(c, dst.data_ptr(), cap, src.data_ptr(), n)
    raw.ZSTDMI_debugReadLzStamps(buf, 1)
    tot = sum(buf[i] for i in range(24))
    print(kind, "level", level, "total wave-leader ticks", tot, flush=True)
    for i in range(24):
        if buf[i]: print(f"   {names[i]:40s} {100.0 * buf[i] / tot:5.1f}%", flush=True)
