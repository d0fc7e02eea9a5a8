"""One-off check on the GPU box: large calls (768 MiB: the fast finder claims its chunks from a counter, levels >= 5 walk a work list
filled by atomics) must give the same bytes call after call, and the bytes must round-trip.  python tools/determinism_check.py"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, numpy as np, datagen
torch.zeros(1, device="cuda")
import zstdsharp_amd as z
lib = z._ffi.load()
for kind, level in (("zipf", 1), ("text", 1), ("mixed", 5), ("mixed", 1)):
    n = 768 << 20
    base = np.frombuffer(datagen.gen(kind, 64 << 20, 3), dtype=np.uint8)
    src = torch.from_numpy(np.tile(base, 12).copy()).cuda()
    cap = lib.ZSTD_compressBound(n)
    outs = []
    for rep in range(3):
        c = lib.ZSTD_createCCtx(); lib.ZSTD_CCtx_setParameter(c, 100, level)
        dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
        cs = lib.ZSTDMI_compressDevice(c, dst.data_ptr(), cap, src.data_ptr(), n)
        assert cs < (1 << 62), cs
        outs.append((cs, dst[:cs].clone())); lib.ZSTD_freeCCtx(c)
    same = all(o[0] == outs[0][0] and bool(torch.equal(o[1], outs[0][1])) for o in outs)
    d = lib.ZSTD_createDCtx(); back = torch.empty(n, dtype=torch.uint8, device="cuda")
    r = lib.ZSTDMI_decompressDevice(d, back.data_ptr(), n, outs[0][1].data_ptr(), outs[0][0])
    print(kind, level, "size", outs[0][0], "deterministic", same, "round trip", r == n and bool(torch.equal(back, src)), flush=True)
    assert same and r == n
