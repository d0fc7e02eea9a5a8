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


def _frames(blob, oracle):
    """[(window the header declares, content size or None, compressed bytes)] of every frame, from the headers alone"""
    out, pos = [], 0
    while pos < len(blob):
        fhd = blob[pos + 4]
        single, fcs_id, did = (fhd >> 5) & 1, fhd >> 6, (0, 1, 2, 4)[fhd & 3]
        p = pos + 5
        window = None
        if not single:
            wl = blob[p]; p += 1
            window = 1 << ((wl >> 3) + 10); window += (window >> 3) * (wl & 7)
        p += did
        fcs = None
        if fcs_id == 0 and single: fcs = blob[p]
        elif fcs_id == 1: fcs = int.from_bytes(blob[p:p + 2], "little") + 256
        elif fcs_id == 2: fcs = int.from_bytes(blob[p:p + 4], "little")
        elif fcs_id == 3: fcs = int.from_bytes(blob[p:p + 8], "little")
        fs = oracle.lib().zso_findFrameCompressedSize(blob[pos:pos + (2 << 20)], min(len(blob) - pos, 2 << 20))
        assert not oracle.is_error(fs)
        out.append((fcs if single else window, fcs, fs))
        pos += fs
    return out


@pytest.mark.parametrize("level", [1, 3, 5])
@pytest.mark.parametrize("windowLog", [10, 11, 12, 15, 16, 17])
def test_window_log_is_honoured_by_every_frame(gpu_lib, oracle, level, windowLog):
    """ZSTD_c_windowLog (U/ZstdCompress.cs:4690-4712, 4817-4929; the window's low limit U/ZstdCompressInternal.cs:787-813): the
    reference never lets an offset or a block exceed 1 << windowLog.  Here: below 16, frames of 1 << windowLog bytes (a frame is its
    own window); 16, independent 64 KiB frames at every level; above, multi-block frames of at most 1 << windowLog of content.  A
    decoder limited to that windowLog (ZSTD_d_windowLogMax) accepts every frame, as T/ZstdNetSteamingTests.cs:290-309 requires."""
    data = datagen.gen("text", 300000, 5) + datagen.gen("zipf", 70001, 6)
    with z.Compressor(level) as c:
        c.SetParameter(101, windowLog)
        assert c.GetParameter(101) == windowLog
        c.SetParameter(201, 1)
        comp = c.Wrap(data)
    fr = _frames(comp, oracle)
    assert max(w for w, _, _ in fr) <= 1 << windowLog, (max(w for w, _, _ in fr), windowLog)
    assert sum(n for _, n, _ in fr) == len(data)
    assert oracle.decompress(comp, len(data)) == data
    with z.Decompressor() as d:
        assert d.Unwrap(comp) == data
    with DecompressionStream(io.BytesIO(comp), 4096) as ds:
        ds.SetParameter(ZSTD_d_windowLogMax, max(windowLog, 10))
        assert ds.ReadToEnd(1 << 16) == data
    if windowLog > 16 and level == 1:
        with z.Compressor(level) as c:
            c.SetParameter(201, 1)
            plain = c.Wrap(data)
        assert len(comp) < len(plain), "a window above 64 KiB must buy something on text at level 1"


@pytest.mark.parametrize("level,n", [(1, 0), (1, 1), (1, 1000), (1, 65536), (1, 200001), (3, 700000), (5, 700000)])
def test_content_size_flag_off_writes_window_descriptor_frames(gpu_lib, oracle, level, n):
    """ZSTD_c_contentSizeFlag = 0 (ZSTD_writeFrameHeader, U/ZstdCompress.cs:4817-4929): no content size in any header, a window
    descriptor instead; the window holds the whole frame (no offset, no block beyond it).  ZSTD_decompressBound is then an upper
    bound (blocks x block size), which is what Unwrap sizes its destination with."""
    data = datagen.gen("text", n, 8)
    with z.Compressor(level) as c:
        c.SetParameter(200, 0)                    # ZSTD_c_contentSizeFlag
        assert c.GetParameter(200) == 0
        comp = c.Wrap(data)
        c.SetParameter(201, 1)
        comp_chk = c.Wrap(data)
    for blob in (comp, comp_chk):
        fr = _frames(blob, oracle)
        assert all(fcs is None for _, fcs, _ in fr), "no frame may carry a content size"
        bound = gpu_lib.ZSTD_decompressBound(blob, len(blob))
        assert n <= bound < (1 << 40)            # blocks x min(window, 128 KiB) per frame (U/ZstdDecompress.cs:971-993)
        assert oracle.decompress(blob, max(bound, 1)) == data
        with z.Decompressor() as d:
            assert d.Unwrap(blob) == data
    assert len(comp_chk) == len(comp) + 4 * len(_frames(comp, oracle))
    # frame sizes: every declared window covers the frame's regenerated size
    pos = 0
    for (w, _, fs) in _frames(comp, oracle):
        piece = oracle.decompress(comp[pos:pos + fs], 1 << 20)
        assert len(piece) <= w, (len(piece), w)
        pos += fs


def test_entropy_stage_bounds_a_poisoned_chunk_record(gpu_lib, oracle):
    """The entropy kernels take sizes from the ChunkMeta the match finder left (nbSeq, litSize, srcSize).  A faulty finder — or an
    experiment build without its emit phase, gpurun_out/call32.log — must not become an out-of-bounds access: every consumer bounds
    the record (meta_checked) and seq_encode bounds its bitstream.  Garbage in, a bounded frame size out, and the context still works."""
    data = datagen.gen("text", 100000, 2)
    with z.Compressor(1) as c:
        for nbSeq, litSize, srcSize, fill in ((16384, 65536, 65536, 0xFF), (0xFFFFFFFF, 5, 65536, 0x00), (100, 0xFFFFFFFF, 65536, 0xA5),
                                              (16384, 0, 65536, 0xFF), (7, 7, 0xFFFFFFFF, 0x5A), (20000, 70000, 65536, 0x11), (16384, 65536, 65536, 0x80)):
            r = gpu_lib.ZSTDMI_debugPoisonedChunk(c.cctx, nbSeq, litSize, srcSize, fill)
            assert not is_error(r), (nbSeq, litSize, srcSize, fill, get_error_code(r))
            assert r <= 65536 + 512, r
        comp = c.Wrap(data)
    assert oracle.decompress(comp, len(data)) == data


def test_target_length_keeps_huffman_literals_on_sparse_input(gpu_lib, oracle):
    """Level 5 + ZSTD_c_targetLength on input without matches: the sparse-input probe must not hand the call to level 1's settings
    WITH the caller's targetLength (at the fast strategy that means raw literals, U/ZstdCompressInternal.cs:146-173): the size stays
    at what Huffman literals give."""
    data = datagen.gen("zipf", 6 << 20, 4)
    with z.Compressor(1) as c:
        l1 = c.Wrap(data)
    with z.Compressor(5) as c:
        c.SetParameter(ZSTD_c_targetLength, 16)
        l5 = c.Wrap(data)
    assert oracle.decompress(l5, len(data)) == data
    assert len(l5) <= len(l1) * 1.02, (len(l5), len(l1))


def test_redundancy_at_a_distance_keeps_the_levels_history(gpu_lib, oracle):
    """The sparse-input probe (levels >= 3 fall back to level 1's path where the finder finds nothing) must also see redundancy
    that only shows at 4 - 64 KiB: a random 16 KiB record repeated repeats nothing inside one 4 KiB tile, yet the level's history
    finds every repeat.  Level 5 by itself must write what it writes with the history forced on, not level 1's frames."""
    import numpy as np
    rng = np.random.default_rng(12)
    recs = [rng.integers(0, 256, 16384, dtype=np.uint8).tobytes() for _ in range(24)]
    order = rng.integers(0, 24, 6 * 64)                      # 6 MiB: every record comes back about 16 times
    data = b"".join(recs[i] + recs[i] for i in order)[:6 << 20]   # (each record twice in a row: a repeat 16 KiB back)
    with z.Compressor(5) as c:
        auto = c.Wrap(data)
        assert gpu_lib.ZSTDMI_CCtx_setHistory(c.cctx, 32 << 10, 0) == 0
        forced = c.Wrap(data)
    with z.Compressor(1) as c:
        l1 = c.Wrap(data)
    assert oracle.decompress(auto, len(data)) == data
    print(f"distance-only redundancy: level 5 {len(auto) / len(data):.4f}, history forced {len(forced) / len(data):.4f}, level 1 {len(l1) / len(data):.4f}")
    assert auto == forced, "the level's own path (history in multi-block frames), not the sparse-input fallback"
    assert auto != l1 and len(auto) <= len(l1) * 1.001, (len(auto), len(l1))


def test_frame_magics_inside_the_payload_do_not_derail_the_parallel_walk(gpu_lib, oracle):
    """The decoder lists a call's frames in parallel (ZSTD_findFrameSizeInfo per 128 KiB segment of the input, U/ZstdDecompress.cs:
    654-760, linked afterwards) and keeps the exact serial walk for what that cannot settle.  Incompressible payload is stored
    verbatim, so it may hold the bytes of a frame header: here a skippable-frame header (U/ZstdDecompress.cs:553-575) whose size
    field reaches far outside its segment — "valid" as far as one segment can tell — every 1000 bytes.  A chain that does not land on
    a frame is not a frame: the call must still take the parallel walk (the serial one costs 2.4 us a frame), and decode exactly."""
    import numpy as np
    rng = np.random.default_rng(5)
    n = 6 << 20
    data = bytearray(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
    for k, off in enumerate(range(16, n - 16, 1000)):
        magic = 0x184D2A50 + (k & 15)
        skip = 150000 + 7919 * (k % 97)                              # out of the segment, inside the input
        data[off:off + 8] = magic.to_bytes(4, "little") + skip.to_bytes(4, "little")
    data = bytes(data)
    with z.Compressor(1) as c, z.Decompressor() as d:
        comp = c.Wrap(data)
        assert (0x184D2A53).to_bytes(4, "little") in comp, "the payload is stored verbatim"
        assert d.Unwrap(comp) == data
        assert gpu_lib.ZSTDMI_debugLastWalkSerial(d.dctx) == 0, "the parallel walk"
        assert oracle.decompress(comp, n) == data
        # ... and what the serial walk is for still gets it: a frame without a content size
        c.SetParameter(200, 0)
        unsized = c.Wrap(data[:300000])
        assert d.Unwrap(unsized) == data[:300000]
        assert gpu_lib.ZSTDMI_debugLastWalkSerial(d.dctx) == 1
