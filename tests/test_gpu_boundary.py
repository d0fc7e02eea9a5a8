"""GPU tests of the drop-in boundary's corner semantics (b): what a call resolves a level to, what
ZSTD_compressCCtx ignores, what the streaming decoder refuses before it allocates.  The checker is the oracle."""
import ctypes
import io

import pytest

import datagen
import zstdsharp_amd as z
from zstdsharp_amd.errors import ZSTD_ErrorCode, ZstdException, get_error_code, is_error
from zstdsharp_amd.streams import DecompressionStream

pytestmark = pytest.mark.gpu

ZSTD_c_targetLength, ZSTD_c_strategy, ZSTD_d_windowLogMax = 106, 107, 100


def _first_block_literals_type(frame: bytes) -> int:
    """literals section type of the first block of a single-segment frame (0 raw, 1 RLE, 2 compressed, 3 treeless)"""
    fhd = frame[4]
    fhs = 5 + (0 if (fhd >> 5) & 1 else 1) + (0, 1, 2, 4)[fhd & 3] + ((1 if (fhd >> 5) & 1 else 0), 2, 4, 8)[fhd >> 6]
    bh = frame[fhs] | (frame[fhs + 1] << 8) | (frame[fhs + 2] << 16)
    assert (bh >> 1) & 3 == 2, "expected a compressed block"
    return frame[fhs + 3] & 3


def test_compressCCtx_ignores_a_loaded_dictionary_and_sticky_parameters(gpu_lib, oracle):
    """ZSTD_compressCCtx = compress_usingDict(NULL, level) (U/ZstdCompress.cs:5751-5776): no dictionary, no checksum, whatever
    the context holds — and the context keeps what it holds for the next ZSTD_compress2."""
    data = datagen.gen("text", 150000, 3)
    dic = datagen.gen("text", 20000, 3)          # shares its vocabulary with the data: a dictionary that really gets used
    with z.Compressor(1) as c:
        c.LoadDictionary(dic)
        c.SetParameter(201, 1)                   # checksum
        with_dict = c.Wrap(data)
        cap = c.GetCompressBound(len(data))
        out = ctypes.create_string_buffer(cap)
        r = gpu_lib.ZSTD_compressCCtx(c.cctx, out, cap, data, len(data), 1)
        assert not is_error(r), get_error_code(r)
        plain = out.raw[:r]
        assert oracle.decompress(plain, len(data)) == data, "must decode WITHOUT the dictionary"
        assert plain[4] & 0x04 == 0, "no checksum flag: default frame parameters"
        assert c.Wrap(data) == with_dict, "sticky parameters and dictionary untouched"
        assert isinstance(oracle.decompress(with_dict, len(data)), int) or oracle.decompress(with_dict, len(data)) != data
        assert oracle.decompress(with_dict, len(data), dic) == data
    with z.Compressor(1) as c2:
        assert c2.Wrap(data) == plain, "same bytes as a context that never saw a dictionary"


def test_negative_levels_store_raw_literals_and_round_trip(gpu_lib, oracle):
    """Negative levels: ZSTD_fast with targetLength = -level (U/ZstdCompress.cs:7915-7920), which switches literal compression
    off (U/ZstdCompressInternal.cs:146-173): literals sections are raw, frames decode under the oracle, sizes do not shrink as
    the level falls."""
    data = datagen.gen("text", 300000, 11)
    sizes = {}
    with z.Decompressor() as d:
        for level in (-1, -5, -20, -131072):
            with z.Compressor(level) as c:
                comp = c.Wrap(data)
            assert oracle.decompress(comp, len(data)) == data, level
            assert d.Unwrap(comp) == data
            assert _first_block_literals_type(comp) == 0, (level, "literals must be stored raw")
            sizes[level] = len(comp)
        with z.Compressor(1) as c:
            l1 = c.Wrap(data)
        assert _first_block_literals_type(l1) == 2
    assert len(l1) < sizes[-1] <= sizes[-5] * 1.01 <= sizes[-20] * 1.02, sizes
    for level in (-5, -20):                      # the oracle's level table reaches row 0: same framing, the reference's parse
        ref = oracle.compress(data, level, 0, 65536)
        print(f"ratio-vs-oracle L{level} text: gpu {sizes[level]} ref {len(ref)} = {sizes[level] / len(ref):.4f}")
        assert sizes[level] <= len(ref) * 1.10, (level, sizes[level], len(ref))
        assert _first_block_literals_type(ref) == 0


def test_explicit_strategy_and_target_length_select_the_finder(gpu_lib, oracle):
    """ZSTD_c_strategy on top of a level (ZSTD_overrideCParams, U/ZstdCompress.cs:2096-2127): level 1 + doubleFast is level 3's
    finder on a 64 KiB chunk, level 1 + greedy is level 5's (given level 5's searchLog); level 1 + targetLength behaves as the negative level of that step."""
    data = datagen.gen("text", 200000, 5)
    def wrap(level, **params):
        with z.Compressor(level) as c:
            for k, v in params.items():
                c.SetParameter({"strategy": ZSTD_c_strategy, "targetLength": ZSTD_c_targetLength, "searchLog": 104}[k], v)
            out = c.Wrap(data)
        assert oracle.decompress(out, len(data)) == data
        return out
    assert wrap(1, strategy=2) == wrap(3)
    # (greedy on top of level 1 keeps level 1's searchLog: the level-5 finder with fewer attempts per position)
    assert len(wrap(1, strategy=3)) < len(wrap(1, strategy=2)) and len(wrap(5)) <= len(wrap(1, strategy=3))
    assert wrap(1, strategy=3, searchLog=5) == wrap(5, searchLog=5)
    assert wrap(3, strategy=1) == wrap(1)
    assert wrap(1, targetLength=5) == wrap(-5)


def _forged_frame(fcs: int, single: bool, window_byte: int = 0) -> bytes:
    """a tiny frame: header + one empty raw last block"""
    if single:
        hdr = bytes([0xE0]) + fcs.to_bytes(8, "little")                      # FHD: 8-byte FCS, single segment
    else:
        hdr = bytes([0xC0, window_byte]) + fcs.to_bytes(8, "little")         # FHD: 8-byte FCS + window descriptor
    return bytes([0x28, 0xB5, 0x2F, 0xFD]) + hdr + bytes([1, 0, 0])


def test_streaming_refuses_oversized_windows_before_allocating(gpu_lib):
    """U/ZstdDecompress.cs:2965-2969: a streamed frame whose window (single segment: its content size) exceeds
    1 << windowLogMax fails with frameParameter_windowTooLarge.  A forged 1 TiB content size must be an error code, not an
    attempt to allocate 1 TiB."""
    for blob in (_forged_frame(1 << 40, True), _forged_frame(0, False, window_byte=(30 - 10) << 3)):
        with pytest.raises(ZstdException) as e:
            DecompressionStream(io.BytesIO(blob)).ReadToEnd()
        assert e.value.Code == ZSTD_ErrorCode.ZSTD_error_frameParameter_windowTooLarge
    # a multi-segment frame that only CLAIMS a huge content size passes the window check; whatever happens next must be an
    # error code (here: allocation refused or corruption), never an abort across the C ABI
    with pytest.raises(ZstdException):
        DecompressionStream(io.BytesIO(_forged_frame(1 << 50, False, window_byte=0))).ReadToEnd()
    # windowLogMax is honoured: a 64 KiB single-segment frame needs windowLog 16
    data = datagen.gen("text", 65536, 1)
    with z.Compressor(1) as c:
        comp = c.Wrap(data)
    with z.Decompressor() as d:
        d.SetParameter(ZSTD_d_windowLogMax, 15)
        with pytest.raises(ZstdException) as e:
            DecompressionStream(io.BytesIO(comp), decompressor=d).ReadToEnd()
        assert e.value.Code == ZSTD_ErrorCode.ZSTD_error_frameParameter_windowTooLarge
        d.SetParameter(ZSTD_d_windowLogMax, 16)
        assert DecompressionStream(io.BytesIO(comp), decompressor=d).ReadToEnd() == data
        assert d.Unwrap(comp) == data            # the one-shot path has no window limit (U/ZstdDecompress.cs:1062-1214)
