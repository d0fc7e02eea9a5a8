"""Frame-walk time of 1 GiB of GPU-built Zipf frames for several seeds: a frame magic that turns up by chance inside the compressed
data may lead a segment's walk astray; the link check then catches it (tools/walk_seeds.py [n seeds])."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen
import zstdsharp_amd as z
lib = z._ffi.load()
n = int(os.environ.get('SIZE_MIB', '1024')) << 20
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    src = torch.from_numpy(datagen.zipf_bytes(n, int(os.environ.get("SEED0", "7")) + seed).copy()).cuda(); torch.cuda.synchronize()
    cap = lib.ZSTD_compressBound(n); dst = torch.empty(cap, dtype=torch.uint8, device="cuda"); back = torch.empty(n, dtype=torch.uint8, device="cuda")
    c, d = z.Compressor(1), z.Decompressor()
    cs = lib.ZSTDMI_compressDevice(c.cctx, dst.data_ptr(), cap, src.data_ptr(), n)
    lib.ZSTDMI_DCtx_setProfiling(d.dctx, 1)
    for _ in range(2): r = lib.ZSTDMI_decompressDevice(d.dctx, back.data_ptr(), n, dst.data_ptr(), cs)
    ms = (ctypes.c_float * 24)(); names = (ctypes.c_char_p * 24)(); k = lib.ZSTDMI_DCtx_getStageTimes(d.dctx, ms, names, 24)
    t = {names[i].decode(): ms[i] for i in range(k)}
    print("seed", int(os.environ.get("SEED0", "7")) + seed, "frame_walk", round(t["frame_walk"], 3), "ms  total decode", round(sum(t.values()), 2), "ok", r == n and bool(torch.equal(back, src)), flush=True)
    c.Dispose(); d.Dispose(); del src, dst, back
