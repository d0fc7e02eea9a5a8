"""Match execution of long frames (decode_origin.hip).  What the reference's Compressor.Wrap writes for any input
is ONE frame (U/ZstdCompress.cs:4690-4815), and ZSTD_execSequence (U/ZstdDecompressBlock.cs:2187-2262) is ordered inside a frame: the
GPU decoder walks a frame on one wave, or — for frames long enough that this walk would dominate — resolves every byte's origin by
pointer jumping.  Both must restore the oracle-built frame bit for bit; the oracle (the reference's algorithm, C) builds the frames."""
import numpy as np
import pytest

import datagen
import zstdsharp_amd as z
from zstdsharp_amd.errors import ZstdException

pytestmark = pytest.mark.gpu


def unwrap(gpu_lib, blob, n, mode, dic=None):
    with z.Decompressor() as d:
        assert gpu_lib.ZSTDMI_DCtx_setLongFrames(d.dctx, mode) == 0
        if dic is not None:
            d.LoadDictionary(dic)
        return d.Unwrap(blob)


@pytest.mark.parametrize("kind,n,level", [("text", 3 << 20, 1), ("text", (3 << 20) + 12345, 5), ("mixed", 5 << 20, 5), ("runs", 2 << 20, 1),
                                          ("period", (2 << 20) + 1, 3), ("zeros", 4 << 20, 1), ("bytei", (1 << 20) + 3, 5), ("zipf", 1 << 20, 1)])
def test_long_frames_by_origin_pointers_match_the_walk(gpu_lib, oracle, kind, n, level):
    """One oracle-built frame of n bytes (128 KiB blocks chained by window, repcodes and repeat-mode tables), decoded both ways.
    `period` and `zeros` are the deep chains: every match copies what the match before it wrote (depth = the number of sequences,
    or of bytes at offset 1) — the pointer jumping needs about log2(depth) rounds for them."""
    data = datagen.gen(kind, n, 21)
    for chk in (0, 1):
        blob = oracle.compress(data, level, chk, 0)
        assert not isinstance(blob, int)
        assert unwrap(gpu_lib, blob, n, 2) == data, "origin path"
        assert unwrap(gpu_lib, blob, n, 1) == data, "one-wave walk"


def test_long_frames_mixed_with_short_ones_and_other_frame_kinds(gpu_lib, oracle):
    """A stream of long and short frames, sized and unsized, raw and RLE blocks among them: only the long ones with sequences take
    the origin path, the rest is walked, in one call."""
    parts = [datagen.gen("text", 2 << 20, 1), datagen.gen("text", 70000, 2), datagen.gen("rand", 1 << 20, 3), datagen.gen("mixed", 3 << 20, 4),
             b"", datagen.gen("zeros", 1 << 20, 5), datagen.gen("text", 1 << 20, 6)]
    blob = b"".join(oracle.compress(p, 5 if i % 2 else 1, i % 2, 0) for i, p in enumerate(parts))
    with z.Compressor(5) as c:                      # GPU-built multi-block frames of 2 MiB without a content size
        assert gpu_lib.ZSTDMI_CCtx_setHistory(c.cctx, 32 << 10, 2 << 20) == 0
        c.SetParameter(200, 0)
        extra = datagen.gen("text", (4 << 20) + 999, 9)
        blob += c.Wrap(extra)
    data = b"".join(parts) + extra
    for mode in (2, 1, 0):
        assert unwrap(gpu_lib, blob, len(data), mode) == data, mode


def test_long_frame_reaching_into_a_dictionary(gpu_lib, oracle):
    """Matches that start in the dictionary (ZSTD_execSequence's extDict branch, U/ZstdDecompressBlock.cs:2223-2250): their first
    bytes are roots in the dictionary, the rest continues at the frame's start."""
    dic = datagen.gen("text", 60000, 31)
    data = dic[-30000:] + datagen.gen("text", 2 << 20, 31) + dic[:5000]
    blob = oracle.compress_dict(data, dic, 1, 1)
    assert not isinstance(blob, int)
    assert oracle.decompress(blob, len(data), dic) == data
    for mode in (2, 1):
        assert unwrap(gpu_lib, blob, len(data), mode, dic) == data


def test_damaged_long_frames_fail_the_same_way(gpu_lib, oracle):
    """Corruption inside a long frame: both executors refuse it (never crash, never return data); an offset beyond the data
    produced so far is corruption_detected (U/ZstdDecompressBlock.cs:2218-2223) on either path."""
    data = datagen.gen("text", 2 << 20, 5)
    blob = bytearray(oracle.compress(data, 5, 1, 0))
    rng = np.random.default_rng(5)
    outcomes = []
    for trial in range(24):
        bad = bytearray(blob)
        for _ in range(1 + trial % 3):
            bad[int(rng.integers(12, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        ref = oracle.decompress(bytes(bad), len(data))
        res = []
        for mode in (2, 1):
            try:
                out = unwrap(gpu_lib, bytes(bad), len(data), mode)
                res.append(("ok", out == data))
            except ZstdException as e:
                res.append(("err", int(e.Code)))
        assert res[0] == res[1], (trial, res)
        if isinstance(ref, int):
            assert res[0][0] == "err", (trial, ref, res)
        else:
            assert res[0] == ("ok", ref == data)
        outcomes.append(res[0][0])
    assert "err" in outcomes


def test_one_256_mib_frame_is_decoded_bit_exactly_and_not_on_one_wave(gpu_lib, oracle):
    """BASELINE configs[4] at its extreme: ONE level-5 frame of 256 MiB (2048 blocks), the shape Compressor.Wrap gives a 256 MiB
    input.  By cost the decoder must send it down the origin path by itself: the one-wave walk takes most of a second."""
    import time
    base = np.frombuffer(datagen.gen("mixed", 64 << 20, 7), dtype=np.uint8)
    data = np.tile(base, 4).tobytes()
    blob = oracle.compress(data, 5, 0, 0)
    assert not isinstance(blob, int) and len(blob) < len(data)
    with z.Decompressor() as d:
        out = d.Unwrap(blob)              # (also the warm-up: workspaces)
        assert out == data
        gpu_lib.ZSTDMI_DCtx_setProfiling(d.dctx, 1)
        t0 = time.perf_counter(); out = d.Unwrap(blob); t = time.perf_counter() - t0
        import ctypes
        ms = (ctypes.c_float * 24)(); names = (ctypes.c_char_p * 24)()
        k = gpu_lib.ZSTDMI_DCtx_getStageTimes(d.dctx, ms, names, 24)
        stages = {names[i].decode(): round(float(ms[i]), 3) for i in range(k)}
    assert out == data
    print(f"one 256 MiB frame: {t * 1e3:.1f} ms through host buffers, stages {stages}")
    assert "origin_jump" in stages, "the frame must take the origin path by cost"
    assert sum(stages.values()) < 100.0, stages          # the one-wave walk alone takes ~700 ms
