"""Development aid: lz stage time and ratio of variant builds of the library (make variant VAR=name DEFS=-D...).
usage: python tools/lz_variant_time.py LEVEL INPUT lib1.so [lib2.so ...]   (one process per library: run via subprocess)"""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 4 or (len(sys.argv) == 4 and not sys.argv[3].endswith(".so")):
    pass
level, kind, libs = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
if len(libs) > 1:
    for l in libs:
        subprocess.run([sys.executable, __file__, str(level), kind, l], check=False)
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes, torch, datagen, numpy as np
import zstdsharp_amd._ffi as ffi
ffi.LIB_PATH = os.path.join(ROOT, "zstdsharp_amd", libs[0])
lib = ffi.load()
n = int(os.environ.get("SIZE_MIB", "256")) << 20
host = np.tile(datagen.text_like(32 << 20, 7), (n + (32 << 20) - 1) // (32 << 20))[:n] if kind == "text" else np.frombuffer(datagen.gen(kind, 32 << 20, 5) * ((n + (32 << 20) - 1) // (32 << 20)), dtype=np.uint8)[:n]
src = torch.from_numpy(host.copy()).cuda(); torch.cuda.synchronize()
cap = lib.ZSTD_compressBound(n); dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
c = lib.ZSTD_createCCtx(); lib.ZSTD_CCtx_setParameter(c, 100, level); lib.ZSTDMI_CCtx_setProfiling(c, 1)
sl = int(os.environ.get("SEARCHLOG", "0"))
if sl: assert lib.ZSTD_CCtx_setParameter(c, 104, sl) == sl
for _ in range(3): cs = lib.ZSTDMI_compressDevice(c, dst.data_ptr(), cap, src.data_ptr(), n)
ms = (ctypes.c_float * 16)(); names = (ctypes.c_char_p * 16)()
k = lib.ZSTDMI_CCtx_getStageTimes(c, ms, names, 16)
st = {names[i].decode(): ms[i] for i in range(k)}
print(f"{libs[0]:40s} searchLog {sl} level {level} {kind}: lz {st.get('lz_fast', 0) * 4:.2f} + region {st.get('lz_region', 0) * 4:.2f} ms/GiB (x 256 MiB / size), all stages ms: { {k: round(v, 3) for k, v in st.items()} }  ratio {cs / n:.5f}", flush=True)
