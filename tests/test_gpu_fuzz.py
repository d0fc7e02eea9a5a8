"""Randomised round trips over the boundary's parameter space (SURVEY.md 8 b): level x windowLog x contentSizeFlag x checksumFlag x
history x dictionary x size x kind of data, seeded.  Every compressed stream must decode under the oracle (the reference's decoder,
restated in C) and under the GPU decoder, on one device and through device workers; sizes must respect ZSTD_compressBound."""
import ctypes

import numpy as np
import pytest

import datagen
import zstdsharp_amd as z
from zstdsharp_amd.errors import get_error_code, is_error

pytestmark = pytest.mark.gpu

KINDS = ["text", "mixed", "zipf", "runs", "period", "rand", "bytei", "zeros"]


@pytest.mark.parametrize("seed", range(120))
def test_random_parameters_round_trip(gpu_lib, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    n = int(rng.choice([0, 1, 7, 255, 256, 4095, 65535, 65536, 65537, 100_000, 262_144, 700_001, 1_500_000, 3_000_017, 5_000_000]))
    level = int(rng.choice([-20, -1, 1, 2, 3, 4, 5, 7, 9, 19]))
    wlog = int(rng.choice([0, 0, 0, 10, 11, 12, 13, 15, 16, 17, 18, 21]))
    csf, chk = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    hist = int(rng.choice([-1, -1, -1, 0, 16 << 10, 32 << 10, 48 << 10]))
    use_dict = rng.integers(0, 4) == 0
    workers = int(rng.choice([1, 1, 2, 3]))
    data = datagen.gen(kind, n, seed)
    dic = None
    if use_dict:
        dic = datagen.gen("text", 30000, seed + 1) if rng.integers(0, 2) else oracle.make_dictionary(datagen.gen(kind, 20000, seed + 2) or b"x" * 64, datagen.gen("text", 50000, seed + 3), 100 + seed)
    desc = dict(kind=kind, n=n, level=level, wlog=wlog, csf=csf, chk=chk, hist=hist, dict=use_dict, workers=workers)
    with z.Compressor(level) as c, z.Decompressor() as d:
        if wlog:
            c.SetParameter(101, wlog)
        c.SetParameter(200, csf); c.SetParameter(201, chk)
        assert gpu_lib.ZSTDMI_CCtx_setHistory(c.cctx, hist, 0) == 0
        if dic is not None:
            c.LoadDictionary(dic); d.LoadDictionary(dic)
        comp = c.Wrap(data)
        assert len(comp) <= gpu_lib.ZSTD_compressBound(n) + 64, desc
        if workers > 1:
            arr = (ctypes.c_int * workers)(*([0] * workers))
            assert not is_error(gpu_lib.ZSTDMI_CCtx_setDevices(c.cctx, arr, workers))
            assert c.Wrap(data) == comp, desc
            assert not is_error(gpu_lib.ZSTDMI_DCtx_setDevices(d.dctx, arr, workers))
        bound = gpu_lib.ZSTD_decompressBound(comp, len(comp))
        assert bound >= n and bound < (1 << 62), desc
        if dic is None or not dic.startswith(bytes([0x37, 0xA4, 0x30, 0xEC])):
            assert oracle.decompress(comp, max(bound, 1), dic) == data, desc       # (the oracle takes raw-content dictionaries)
        assert d.Unwrap(comp) == data, desc
