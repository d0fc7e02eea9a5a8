import os, sys, ctypes
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import torch
torch.cuda.init()
import zstdsharp_amd as z
from bench import make_zipf, stage_times
lib = z._ffi.load()
dev = torch.device("cuda", 0)
n = 1 << 30
src = make_zipf(n, 1234, dev); torch.cuda.synchronize()
cap = lib.ZSTD_compressBound(n)
comp = torch.empty(cap + 8192, dtype=torch.uint8, device=dev)
c = z.Compressor(1)
cs = lib.ZSTDMI_compressDevice(c.cctx, comp.data_ptr(), cap, src.data_ptr(), n)
def run(d, inp, out, reps=3):
    t = []
    for _ in range(reps):
        r = lib.ZSTDMI_decompressDevice(d.dctx, out.data_ptr(), n, inp.data_ptr(), cs); assert r == n
        t.append(stage_times(lib, d.dctx, lib.ZSTDMI_DCtx_getStageTimes)["decode_literals"])
    return min(t)
back = torch.empty(n, dtype=torch.uint8, device=dev)
print("A: new DCtx (new scratch) each time, same torch buffers")
for i in range(5):
    d = z.Decompressor(); lib.ZSTDMI_DCtx_setProfiling(d.dctx, 1)
    print("  %.3f" % run(d, comp, back)); d.Dispose()
print("B: same DCtx, new output buffer each time")
d = z.Decompressor(); lib.ZSTDMI_DCtx_setProfiling(d.dctx, 1)
keep = []
for i in range(4):
    out = torch.empty(n, dtype=torch.uint8, device=dev); keep.append(out)
    print("  %.3f  out%%2MiB=%d" % (run(d, comp, out), out.data_ptr() % (2 << 20)))
print("C: same DCtx, compressed input copied to a new buffer each time")
for i in range(4):
    inp = comp.clone(); keep.append(inp)
    print("  %.3f  in%%2MiB=%d" % (run(d, inp, back), inp.data_ptr() % (2 << 20)))
