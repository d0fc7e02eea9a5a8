"""Diagnostic (stamped build: make -C zstdsharp_amd/csrc stamps): where exec_matches spends its time on oracle-built frames.
python tools/exec_stamps.py [kind] [frame MiB] [total MiB] [level]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np, datagen
torch.zeros(1, device="cuda")
import zstdsharp_amd._ffi as ffi
ffi.LIB_PATH = os.path.join(ROOT, "zstdsharp_amd", "libzstd_mi355x_stamps.so")
lib = ffi.load()
raw = ctypes.CDLL(ffi.LIB_PATH); raw.ZSTDMI_debugReadSeqStamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
import oracle_lib as o
from concurrent.futures import ThreadPoolExecutor
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
d = lib.ZSTD_createDCtx(); lib.ZSTDMI_DCtx_setProfiling(d, 1); lib.ZSTDMI_DCtx_setLongFrames(d, 1)
buf = (ctypes.c_ulonglong * 16)()
for _ in range(2):
    raw.ZSTDMI_debugReadSeqStamps(buf, 1)
    r = lib.ZSTDMI_decompressDevice(d, out.data_ptr(), n, comp.data_ptr(), comp.numel()); assert r == n
raw.ZSTDMI_debugReadSeqStamps(buf, 0)
ms = (ctypes.c_float * 24)(); names = (ctypes.c_char_p * 24)()
k = lib.ZSTDMI_DCtx_getStageTimes(d, ms, names, 24)
print({names[i].decode(): round(ms[i], 3) for i in range(k)})
e = [buf[8 + i] for i in range(8)]
frames = n // fb
print(f"exec: per frame-wave ticks: load/unpack {e[0] // frames}, dependency analysis {e[1] // frames}, rounds {e[2] // frames}; batches/frame {e[7] / frames:.1f}, rounds/batch {e[6] / max(e[7], 1):.2f}, ticks/batch {(e[0] + e[1] + e[2]) / max(e[7], 1):.0f}")
