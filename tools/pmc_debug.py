import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, zstdsharp_amd as z, oracle_lib as o, datagen
lib = z._ffi.load()
n = 64 << 20
src = torch.from_numpy(datagen.zipf_bytes(n, 5)).cuda()
cap = lib.ZSTD_compressBound(n); dst = torch.empty(cap, dtype=torch.uint8, device="cuda"); back = torch.empty(n, dtype=torch.uint8, device="cuda")
c, d = z.Compressor(1), z.Decompressor()
for it in range(3):
    cs = lib.ZSTDMI_compressDevice(c.cctx, dst.data_ptr(), cap, src.data_ptr(), n)
    comp = dst[:cs].cpu().numpy().tobytes() if cs < (1 << 62) else b""
    bound = lib.ZSTD_decompressBound(comp, len(comp))
    r = o.decompress(comp, n)
    print("iter", it, "cs", cs, "ratio", cs / n, "bound", bound, "oracle ok", (r == src.cpu().numpy().tobytes()) if isinstance(r, bytes) else r, flush=True)
    rr = lib.ZSTDMI_decompressDevice(d.dctx, back.data_ptr(), n, dst.data_ptr(), cs)
    print("   gpu dec", rr if rr < (1 << 62) else lib.ZSTD_getErrorName(rr), bool(torch.equal(src, back)), flush=True)
