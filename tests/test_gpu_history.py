"""GPU tests of cross-chunk history: blocks that match into the input in front of them, grouped into multi-block
frames — the window ZSTD_compress_frameChunk's block loop carries (U/ZstdCompress.cs:4705-4807).  The checker is the oracle's
decoder; the yardstick for size is the oracle's encoder on the same data with the same frame size."""
import pytest

import datagen
import zstdsharp_amd as z
from zstdsharp_amd.errors import is_error

pytestmark = pytest.mark.gpu

ZSTD_c_windowLog, ZSTD_c_checksumFlag = 101, 201


def walk_frames(lib, comp: bytes):
    """[(frame content size, [(block type, last, block size)...])] of a concatenation of single-segment frames"""
    out, pos = [], 0
    while pos < len(comp):
        fsz = lib.ZSTD_findFrameCompressedSize(comp[pos:], len(comp) - pos)
        assert not is_error(fsz)
        f = comp[pos:pos + fsz]
        fhd = f[4]
        assert (fhd >> 5) & 1, "single segment"
        did = (0, 1, 2, 4)[fhd & 3]
        fcs_bytes = (1, 2, 4, 8)[fhd >> 6]
        fcs = int.from_bytes(f[5 + did:5 + did + fcs_bytes], "little") + (256 if fcs_bytes == 2 else 0)
        p, blocks = 5 + did + fcs_bytes, []
        while True:
            bh = int.from_bytes(f[p:p + 3], "little")
            last, btype, bsz = bh & 1, (bh >> 1) & 3, bh >> 3
            blocks.append((btype, last, bsz))
            p += 3 + (1 if btype == 1 else bsz)
            if last:
                break
        assert p + (4 if fhd & 4 else 0) == fsz
        out.append((fcs, blocks))
        pos += fsz
    return out


def set_history(lib, c, hist, frame=0):
    assert lib.ZSTDMI_CCtx_setHistory(c.cctx, hist, frame) == 0


@pytest.mark.parametrize("level", [1, 3, 5])
@pytest.mark.parametrize("kind,n", [("text", 700001), ("mixed", 1 << 20), ("runs", 300000), ("zipf", 262144 + 5), ("period", 262144), ("rand", 200000),
                                    ("text", 65537), ("text", 32768 * 9)])
def test_history_frames_round_trip_under_both_decoders(gpu_lib, oracle, level, kind, n):
    data = datagen.gen(kind, n, 21)
    with z.Compressor(level) as c, z.Decompressor() as d:
        set_history(gpu_lib, c, 32 << 10)
        for chk in (0, 1):
            c.SetParameter(ZSTD_c_checksumFlag, chk)
            comp = c.Wrap(data)
            assert oracle.decompress(comp, n) == data, "the reference's decoder (oracle) must restore the input"
            assert d.Unwrap(comp) == data
            frames = walk_frames(gpu_lib, comp)
            assert sum(f[0] for f in frames) == n
            # 256 KiB of content per frame in 32 KiB blocks; only the last block of a frame says so
            assert all(f[0] == (256 << 10) for f in frames[:-1])
            for fcs, blocks in frames:
                assert len(blocks) == (fcs + 32767) // 32768
                assert [b[1] for b in blocks] == [0] * (len(blocks) - 1) + [1]


def test_history_buys_ratio_on_text_and_stays_near_the_oracle_at_the_same_frame_size(gpu_lib, oracle):
    """With 32 KiB of the input in front of every block as history the frames shrink (matches reach back 32-64 KiB instead of
    0-64 KiB); 48 KiB blocks with 16 KiB of history land in between.  Yardstick: the oracle's encoder with 256 KiB frames, whose
    window is the whole frame (what in-LDS history cannot reach): the GPU stays within 9 % of it at level 1."""
    data = datagen.gen("text", 2 << 20, 4)
    sizes = {}
    with z.Compressor(1) as c:
        for hist in (0, 16 << 10, 32 << 10):
            set_history(gpu_lib, c, hist)
            sizes[hist] = len(c.Wrap(data))
    assert sizes[32 << 10] < sizes[16 << 10] < sizes[0]
    assert sizes[32 << 10] <= 0.985 * sizes[0]
    ref = len(oracle.compress(data, 1, 0, 256 << 10))
    assert sizes[32 << 10] <= 1.09 * ref, (sizes, ref)


def test_history_is_on_by_level_and_by_window_log_and_off_with_a_dictionary(gpu_lib, oracle):
    data = datagen.gen("text", 600000, 8)

    def frame_sizes(c):
        return [f[0] for f in walk_frames(gpu_lib, c.Wrap(data))]

    with z.Compressor(1) as c:
        assert frame_sizes(c)[0] == 65536, "level 1 (fast strategy): independent 64 KiB frames, the throughput configuration"
        c.SetParameter(ZSTD_c_windowLog, 18)
        assert frame_sizes(c)[0] == (256 << 10), "a window above 64 KiB was asked for"
        c.SetParameter(ZSTD_c_windowLog, 16)
        assert frame_sizes(c)[0] == 65536
    with z.Compressor(3) as c:
        assert frame_sizes(c)[0] == (256 << 10), "strategies above fast (levels >= 3) carry history by default"
        set_history(gpu_lib, c, 0)
        assert frame_sizes(c)[0] == 65536
        set_history(gpu_lib, c, -1)
        dic = datagen.gen("text", 20000, 8)
        c.LoadDictionary(dic)
        comp = c.Wrap(data)
        assert oracle.decompress(comp, len(data), dic) == data
        assert all(len(f[1]) == 1 for f in walk_frames(gpu_lib, comp)), "with a dictionary every chunk is its own frame behind the dictionary"


def test_history_frame_size_is_settable_and_frames_never_straddle_passes(gpu_lib, oracle):
    data = datagen.gen("text", (1 << 20) + 777, 9)
    with z.Compressor(3) as c, z.Decompressor() as d:
        set_history(gpu_lib, c, 32 << 10, 1 << 20)
        assert gpu_lib.ZSTDMI_CCtx_setPassChunks(c.cctx, 37) == 0          # not a multiple of the 32 blocks of a frame
        comp = c.Wrap(data)
        frames = walk_frames(gpu_lib, comp)
        assert [f[0] for f in frames] == [1 << 20, 777]
        assert oracle.decompress(comp, len(data)) == data and d.Unwrap(comp) == data
    with z.Compressor(1) as c:
        assert is_error(gpu_lib.ZSTDMI_CCtx_setHistory(c.cctx, 49 << 10, 0))
        assert is_error(gpu_lib.ZSTDMI_CCtx_setHistory(c.cctx, 0, 1000))
