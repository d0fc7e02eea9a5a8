#!/usr/bin/env python3
"""One-off randomized differential sweep on the GPU box (not part of the test suite; run through gpurun):
random sizes / data kinds / levels / checksum / dictionary kind -> GPU compress must decode under the oracle and the GPU,
oracle frames must decode under the GPU.  Prints a summary line; exits non-zero on the first mismatch."""
import os, sys, random, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
torch.cuda.init()
import datagen, oracle_lib as oracle
import zstdsharp_amd as z

def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    rng = random.Random(seed)
    kinds = datagen.KINDS
    content = datagen.gen("text", 30000, 5); sample = datagen.gen("text", 60000, 6)
    dicts = {"none": None, "raw": content, "fmt": oracle.make_dictionary(content, sample, 0xABCDEF)}
    t0 = time.time(); n_cases = 0
    comps = {}; decs = {}
    while time.time() - t0 < budget:
        kind = rng.choice(kinds); level = rng.choice([1, 1, 3, 5, 5, 7, 9]); chk = rng.randrange(2); dk = rng.choice(list(dicts))
        n = rng.choice([rng.randrange(0, 300), rng.randrange(0, 70000), rng.randrange(0, 400000), 65536 * rng.randrange(1, 5) + rng.randrange(-2, 3),
                        32768 * rng.randrange(1, 40) + rng.randrange(-2, 3)])
        if rng.randrange(200) == 0: n = rng.choice([4 << 20, (5 << 20) + 123, (4 << 20) + 65536 * 3 + 1])      # (calls large enough for the levels >= 3 sparse-input probe)
        data = datagen.gen(kind, n, rng.randrange(1 << 30))
        # cross-chunk history (row f-1): by level, off, or 16/32/48 KiB with frames of 128 KiB .. 1 MiB
        hist = rng.choice([(-1, 0), (-1, 0), (0, 0), (16 << 10, 128 << 10), (32 << 10, 256 << 10), (48 << 10, 1 << 20), (32 << 10, 64 << 10)])
        key = (level, chk, dk, hist)
        if key not in comps:
            c = z.Compressor(level); c.SetParameter(201, chk); c.LoadDictionary(dicts[dk]); comps[key] = c
            assert z._ffi.load().ZSTDMI_CCtx_setHistory(c.cctx, hist[0], hist[1]) == 0
        if dk not in decs:
            d = z.Decompressor(); d.LoadDictionary(dicts[dk]); decs[dk] = d
        c, d = comps[key], decs[dk]
        if rng.randrange(4) == 0:
            assert z._ffi.load().ZSTDMI_DCtx_setLiteralDecoder(d.dctx, rng.randrange(4)) == 0
        comp = c.Wrap(data)
        want = oracle.decompress(comp, n, dicts[dk])
        assert want == data, ("oracle rejects GPU frame", kind, n, level, chk, dk, want if isinstance(want, int) else "mismatch")
        assert d.Unwrap(comp) == data, ("GPU round trip", kind, n, level, chk, dk)
        if level == 1 or dk == "none":
            ref = oracle.compress_dict(data, dicts[dk], level, chk) if dk != "none" else oracle.compress(data, level, chk, rng.choice([0, 65536]))
            if not isinstance(ref, int):
                assert d.Unwrap(ref) == data, ("GPU decode of oracle frame", kind, n, level, chk, dk)
        n_cases += 1
    print(f"stress sweep seed {seed}: {n_cases} cases in {time.time() - t0:.0f} s, all consistent")

if __name__ == "__main__":
    main()
