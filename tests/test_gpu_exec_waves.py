"""Match execution with several waves per frame (decode_seq.hip exec_matches_wide_kernel).  ZSTD_execSequence
(U/ZstdDecompressBlock.cs:2187-2262) is ordered inside a frame; the GPU runs a batch of 64 x W sequences per round of dependent copies
on a W-wave workgroup (W by the number of frames in the call, or set).  Every W must restore oracle-built frames bit for bit."""
import numpy as np
import pytest

import datagen
import zstdsharp_amd as z
from zstdsharp_amd.errors import ZstdException

pytestmark = pytest.mark.gpu

WAVES = (1, 2, 4, 8, 16)


def unwrap(gpu_lib, blob, waves, dic=None):
    with z.Decompressor() as d:
        assert gpu_lib.ZSTDMI_DCtx_setLongFrames(d.dctx, 1) == 0          # the walk, not the origin pointers
        assert gpu_lib.ZSTDMI_DCtx_setExecWaves(d.dctx, waves) == 0
        if dic is not None:
            d.LoadDictionary(dic)
        return d.Unwrap(blob)


@pytest.mark.parametrize("kind,n,level", [("text", (1 << 20) + 777, 5), ("text", 300000, 1), ("mixed", 2 << 20, 5), ("runs", 1 << 20, 1),
                                          ("period", (1 << 20) + 1, 3), ("zeros", 1 << 20, 1), ("bytei", 200003, 5), ("zipf", 1 << 18, 1)])
def test_every_width_restores_oracle_frames(gpu_lib, oracle, kind, n, level):
    """One oracle-built frame (128 KiB blocks chained by window and repcodes), with and without a checksum.  `period`, `zeros` and
    `runs` are the deep chains and the long matches: every match of a batch reads what the one before it wrote."""
    data = datagen.gen(kind, n, 33)
    for chk in (0, 1):
        blob = oracle.compress(data, level, chk, 0)
        assert not isinstance(blob, int)
        for w in WAVES:
            assert unwrap(gpu_lib, blob, w) == data, (w, chk)


def test_many_frames_of_unequal_size_in_one_call(gpu_lib, oracle):
    """Frames of 1 byte to 1 MiB, raw and RLE blocks and empty frames among them, at every width and at the width the call picks."""
    rng = np.random.default_rng(7)
    parts = []
    for i in range(40):
        kind = ("text", "mixed", "runs", "rand", "zeros", "period")[i % 6]
        parts.append(datagen.gen(kind, int(rng.integers(1, 1 << (10 + i % 11))), 100 + i))
    parts[5] = b""
    # (level 5 on the larger pieces only: the oracle holds the level's parameters for inputs above 16 KiB)
    frames = [oracle.compress(p, (1, 3, 5 if len(p) > (128 << 10) else 2)[i % 3], i % 2, 0) for i, p in enumerate(parts)]
    assert not any(isinstance(f, int) for f in frames), [f for f in frames if isinstance(f, int)]
    blob = b"".join(frames)
    data = b"".join(parts)
    for w in WAVES + (0,):
        assert unwrap(gpu_lib, blob, w) == data, w


def test_matches_reaching_into_a_dictionary(gpu_lib, oracle):
    """ZSTD_execSequence's extDict branch (U/ZstdDecompressBlock.cs:2223-2250) on every width."""
    dic = datagen.gen("text", 60000, 41)
    data = dic[-30000:] + datagen.gen("text", 1 << 20, 41) + dic[:5000]
    blob = oracle.compress_dict(data, dic, 1, 1)
    assert not isinstance(blob, int)
    for w in WAVES:
        assert unwrap(gpu_lib, blob, w, dic) == data, w


def test_damage_is_refused_the_same_way_at_every_width(gpu_lib, oracle):
    """Bit flips inside a frame: every width gives what the one-wave walk gives — the same error or (a flip the format does not see,
    no checksum) the same bytes; an offset beyond the data produced so far is corruption_detected (:2218-2223)."""
    data = datagen.gen("text", 1 << 20, 6)
    blob = bytearray(oracle.compress(data, 5, 0, 0))
    rng = np.random.default_rng(6)
    for trial in range(16):
        bad = bytearray(blob)
        for _ in range(1 + trial % 3):
            bad[int(rng.integers(12, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        res = []
        for w in WAVES:
            try:
                res.append(("ok", unwrap(gpu_lib, bytes(bad), w)))
            except ZstdException as e:
                res.append(("err", int(e.Code)))
        assert all(r == res[0] for r in res), (trial, [r[0] if r[0] == "ok" else r for r in res])


def test_width_is_validated(gpu_lib):
    with z.Decompressor() as d:
        assert gpu_lib.ZSTDMI_DCtx_setExecWaves(d.dctx, 3) != 0
        assert gpu_lib.ZSTDMI_DCtx_setExecWaves(d.dctx, 32) != 0
        assert gpu_lib.ZSTDMI_DCtx_setExecWaves(d.dctx, 0) == 0
