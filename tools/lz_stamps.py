"""Diagnostic: phase shares of lz_fast_kernel from the stamped build (make -C zstdsharp_amd/csrc stamps)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen, numpy as np
import zstdsharp_amd._ffi as ffi
ffi.LIB_PATH = os.path.join(ROOT, "zstdsharp_amd", "libzstd_mi355x_stamps.so")
lib = ffi.load()
raw = ctypes.CDLL(ffi.LIB_PATH); raw.ZSTDMI_debugReadLzStamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
raw.ZSTDMI_debugReadHufStamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
raw.ZSTDMI_debugReadSeqStamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
raw.ZSTDMI_debugReadSeqEncStamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
enames = ["repcodes", "histograms", "tables (lane 0)", "pack+flush (+batch load)", "state chains"]
snames = ["other", "state chain", "fields+reps", "literals", "indep matches", "dependent matches"]
hnames = ["hist", "decide+place", "bucket sort", "build tree", "depths", "maxHeight", "codes+bits", "weights+final"]
names = ["stage", "probe(pre-A)", "barrier A", "verify(phase B)", "barrier B", "emit(+sparse)", "barrier C", "literals", "dense tile: select", "dense tile: finish+rank", "region: candidates (I)", "region: load pass", "region: speculative parse", "region: real parse", "region: links+scans", "region: emit+literals"]
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
for kind in ("zipf", "text"):
    host = datagen.zipf_bytes(n, 3) if kind == "zipf" else np.tile(datagen.text_like(32 << 20, 7), (n + (32 << 20) - 1) // (32 << 20))[:n]
    src = torch.from_numpy(host.copy()).cuda(); torch.cuda.synchronize()
    cap = lib.ZSTD_compressBound(n); dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    c = lib.ZSTD_createCCtx(); lib.ZSTD_CCtx_setParameter(c, 100, 1)
    lib.ZSTDMI_compressDevice(c, dst.data_ptr(), cap, src.data_ptr(), n)
    buf = (ctypes.c_ulonglong * 24)(); raw.ZSTDMI_debugReadLzStamps(buf, 1); raw.ZSTDMI_debugReadHufStamps(buf, 1); raw.ZSTDMI_debugReadSeqEncStamps(buf, 1)
    lib.ZSTDMI_compressDevice(c, dst.data_ptr(), cap, src.data_ptr(), n)
    raw.ZSTDMI_debugReadLzStamps(buf, 1)
    tot = sum(buf[i] for i in range(16)); chunks = n // 65536
    print(kind, "lz wave-cycles/chunk/16", tot // chunks // 16, {names[i]: f"{100.0 * buf[i] / tot:.1f}%" for i in range(16)}, flush=True)
    raw.ZSTDMI_debugReadHufStamps(buf, 1)
    tot = sum(buf[i] for i in range(8))
    print(kind, "huf_build cycles/chunk", tot // chunks, {hnames[i]: f"{100.0 * buf[i] / tot:.1f}%" for i in range(8)}, flush=True)
    raw.ZSTDMI_debugReadSeqEncStamps(buf, 1)
    tot = sum(buf[i] for i in range(5))
    print(kind, "seq_encode cycles/chunk", tot // chunks, {enames[i]: f"{100.0 * buf[i] / tot:.1f}%" for i in range(5)}, flush=True)
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    cs = lib.ZSTDMI_compressDevice(c, dst.data_ptr(), cap, src.data_ptr(), n)
    d = lib.ZSTD_createDCtx()
    lib.ZSTDMI_decompressDevice(d, back.data_ptr(), n, dst.data_ptr(), cs)
    raw.ZSTDMI_debugReadSeqStamps(buf, 1)
    r = lib.ZSTDMI_decompressDevice(d, back.data_ptr(), n, dst.data_ptr(), cs)
    raw.ZSTDMI_debugReadSeqStamps(buf, 1)
    tot = sum(buf[i] for i in range(6))
    print(kind, "decode_sequences cycles/chunk", tot // chunks, {snames[i]: f"{100.0 * buf[i] / tot:.1f}%" for i in range(6)},
          "dependent matches per batch", round(buf[6] / max(buf[7], 1), 2), "batches/chunk", round(buf[7] / chunks, 1), "ok", r == n, flush=True)
    print(kind, "literal streams: passes/stream", round(buf[8] / max(buf[9], 1), 2), "sync cycles/stream", buf[10] // max(buf[9], 1), "write cycles/stream", buf[11] // max(buf[9], 1), flush=True)
