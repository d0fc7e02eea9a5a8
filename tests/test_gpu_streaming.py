"""the streaming adapters (ZSTD_compressStream2 / ZSTD_decompressStream behind mirrors of
CompressionStream / DecompressionStream).  Cases follow T/ZstdNetSteamingTests.cs; the checker for every produced stream
is the oracle decoder (tests only)."""
import io

import pytest

import datagen
import zstdsharp_amd as z
from zstdsharp_amd.streams import CompressionStream, DecompressionStream, EndOfStreamException

pytestmark = pytest.mark.gpu

LARGE = 1024 * 1024                       # DataGenerator.LargeBufferSize (T/ZstdNetSteamingTests.cs:22)


def sequential(n):                        # DataFill.Sequential (T/ZstdNetSteamingTests.cs:37-41)
    return bytes(i % 256 for i in range(256)) * (n // 256) + bytes(i % 256 for i in range(n % 256))


def test_streaming_compression_zero_and_one_byte(gpu_lib, oracle):
    """T/ZstdNetSteamingTests.cs:49-87."""
    data = bytes([0, 0, 0, 1, 2, 3, 4, 0, 0, 0])
    tmp = io.BytesIO()
    with CompressionStream(tmp) as cs:
        cs.Write(data, 0, 0); cs.Write(b"")
        cs.Write(data, 3, 1); cs.Write(data[4:5]); cs.Flush()
        cs.Write(data, 5, 1); cs.Write(data[6:7]); cs.Flush()
    blob = tmp.getvalue()
    assert oracle.decompress(blob, 64) == data[3:7]
    tmp.seek(0)
    result = bytearray(len(data))
    with DecompressionStream(tmp) as ds:
        assert ds.Read(0) == b""
        for i in (3, 4, 5, 6):
            b = ds.Read(1); assert len(b) == 1; result[i] = b[0]
    assert bytes(result) == data


@pytest.mark.parametrize("data,offset,count", [(b"", 0, 0), (b"\x01\x02\x03", 1, 2), (b"\x01\x02\x03", 0, 2), (b"\x01\x02\x03", 1, 1), (b"\x01\x02\x03", 0, 3)])
def test_streaming_compression_simple_write(gpu_lib, oracle, data, offset, count):
    """T/ZstdNetSteamingTests.cs:90-112 (an empty stream still yields a valid, empty frame)."""
    tmp = io.BytesIO()
    with CompressionStream(tmp) as cs:
        cs.Write(data, offset, count)
    blob = tmp.getvalue()
    assert len(blob) > 0
    assert oracle.decompress(blob, 16) == data[offset:offset + count]
    tmp.seek(0)
    with DecompressionStream(tmp) as ds:
        assert ds.ReadToEnd() == data[offset:offset + count]


@pytest.mark.parametrize("readCount", [1, 2, 3, 5, 9, 10])
def test_streaming_decompression_simple_read(gpu_lib, readCount):
    """T/ZstdNetSteamingTests.cs:114-147."""
    data = bytes(range(10))
    tmp = io.BytesIO()
    with CompressionStream(tmp) as cs:
        cs.Write(data)
    tmp.seek(0)
    got = bytearray()
    with DecompressionStream(tmp) as ds:
        while True:
            b = ds.Read(min(readCount, len(data) - len(got)))
            if not b:
                break
            assert len(b) <= readCount
            got += b
    assert bytes(got) == data


def test_streaming_decompression_truncated_input(gpu_lib):
    """T/ZstdNetSteamingTests.cs:149-168: a cut stream ends in EndOfStreamException, not in silence."""
    tmp = io.BytesIO()
    with CompressionStream(tmp) as cs:
        cs.Write(sequential(LARGE))
    blob = tmp.getvalue()
    cut = io.BytesIO(blob[:min(32, len(blob) // 3)])
    with pytest.raises(EndOfStreamException):
        with DecompressionStream(cut) as ds:
            ds.ReadToEnd()
    # cut in the middle of a later frame: everything before it is delivered, then the exception
    cut = io.BytesIO(blob[:len(blob) - 5])
    with pytest.raises(EndOfStreamException):
        with DecompressionStream(cut) as ds:
            ds.ReadToEnd()


def test_flush_makes_data_decodable(gpu_lib, oracle):
    """T/ZstdNetSteamingTests.cs:170-189."""
    tmp = io.BytesIO()
    cs = CompressionStream(tmp)
    cs.Write(b"\x00"); cs.Flush()
    assert tmp.tell() > 0
    assert oracle.decompress(tmp.getvalue(), 8) == b"\x00"
    with DecompressionStream(io.BytesIO(tmp.getvalue())) as ds:
        assert ds.ReadToEnd() == b"\x00"
    cs.Dispose()


def test_round_trip_batch_to_streaming_and_back(gpu_lib, oracle):
    """T/ZstdNetSteamingTests.cs:225-267: Wrap -> DecompressionStream, CompressionStream -> Unwrap, and shrinkage."""
    data = sequential(LARGE)
    with z.Compressor() as c:
        comp = c.Wrap(data)
    with DecompressionStream(io.BytesIO(comp)) as ds:
        assert ds.ReadToEnd() == data
    tmp = io.BytesIO()
    with CompressionStream(tmp) as cs:
        cs.Write(data)
    assert tmp.tell() < len(data)
    with z.Decompressor() as d:
        assert d.Unwrap(tmp.getvalue()) == data
    assert oracle.decompress(tmp.getvalue(), len(data)) == data


def frame_windows(blob):
    """(window, content size or None) of every frame of a stream, from the headers alone (U/ZstdDecompress.cs:462-634)."""
    import oracle_lib as o
    out, pos = [], 0
    while pos < len(blob):
        fhd = blob[pos + 4]
        single, fcs_id, did = (fhd >> 5) & 1, fhd >> 6, (0, 1, 2, 4)[fhd & 3]
        p = pos + 5
        window = None
        if not single:
            wl = blob[p]; p += 1
            window = (1 << ((wl >> 3) + 10)); window += (window >> 3) * (wl & 7)
        p += did
        fcs = None
        if fcs_id == 0 and single: fcs = blob[p]
        elif fcs_id == 1: fcs = int.from_bytes(blob[p:p + 2], "little") + 256
        elif fcs_id == 2: fcs = int.from_bytes(blob[p:p + 4], "little")
        elif fcs_id == 3: fcs = int.from_bytes(blob[p:p + 8], "little")
        out.append((fcs if single else window, fcs))
        pos += o.lib().zso_findFrameCompressedSize(blob[pos:pos + (1 << 20)], min(len(blob) - pos, 1 << 20))
    return out


@pytest.mark.parametrize("useDict", [False, True])
@pytest.mark.parametrize("advanced", [False, True])
@pytest.mark.parametrize("zstdBufferSize", [1, 7, 1024, 65535, LARGE + 1])
@pytest.mark.parametrize("copyBufferSize", [101, 65535, LARGE + 1])
def test_round_trip_streaming_to_streaming(gpu_lib, oracle, useDict, advanced, zstdBufferSize, copyBufferSize):
    """T/ZstdNetSteamingTests.cs:269-318, every axis: dictionary (a formatted one; the reference trains its own with ZDICT, which
    is outside the path), advanced = ZSTD_c_windowLog 11 + checksum on the compression stream and ZSTD_d_windowLogMax 11 on the
    decompression stream — so no frame may declare a window above 2 KiB.  (Without the smallest copy sizes on the full 1 MiB:
    python loop time; the small ones run on 50 000 bytes below.)"""
    data = sequential(LARGE)
    dic = oracle.make_dictionary(datagen.gen("text", 20000, 3) + sequential(4096), datagen.gen("text", 60000, 4), 4242) if useDict else None
    tmp = io.BytesIO()
    with CompressionStream(tmp, 3, zstdBufferSize) as cs:
        cs.LoadDictionary(dic)
        if advanced:
            cs.SetParameter(101, 11)                    # ZSTD_c_windowLog
            cs.SetParameter(201, 1)                     # ZSTD_c_checksumFlag
        for lo in range(0, len(data), copyBufferSize):
            cs.Write(data, lo, min(copyBufferSize, len(data) - lo))
    blob = tmp.getvalue()
    if advanced:
        fw = frame_windows(blob)
        assert max(w for w, _ in fw) <= 2048 and sum(c for _, c in fw) == len(data)
    if not useDict:
        assert oracle.decompress(blob, len(data)) == data
    tmp.seek(0)
    out = bytearray()
    with DecompressionStream(tmp, zstdBufferSize if zstdBufferSize >= 7 else 7) as ds:
        ds.LoadDictionary(dic)
        if advanced:
            ds.SetParameter(100, 11)                    # ZSTD_d_windowLogMax
        while True:
            b = ds.Read(copyBufferSize)
            if not b:
                break
            out += b
    assert bytes(out) == data


@pytest.mark.parametrize("copyBufferSize", [1, 2, 7])
def test_round_trip_tiny_copies(gpu_lib, copyBufferSize):
    data = datagen.gen("text", 50000, 5)
    tmp = io.BytesIO()
    with CompressionStream(tmp, 1, 7) as cs:
        for lo in range(0, len(data), copyBufferSize):
            cs.Write(data, lo, min(copyBufferSize, len(data) - lo))
    tmp.seek(0)
    out = bytearray()
    with DecompressionStream(tmp, 7) as ds:
        while True:
            b = ds.Read(copyBufferSize)
            if not b:
                break
            out += b
    assert bytes(out) == data


def test_large_stream_crosses_batches(gpu_lib, oracle):
    """More than one 16 MiB batch through e_continue, mixed data, then GPU and oracle decode."""
    data = datagen.gen("mixed", 40 * 1024 * 1024 + 12345, 9)
    tmp = io.BytesIO()
    with CompressionStream(tmp, 1) as cs:
        for lo in range(0, len(data), 3_000_000):
            cs.Write(data, lo, min(3_000_000, len(data) - lo))
    blob = tmp.getvalue()
    with z.Decompressor() as d:
        assert d.Unwrap(blob) == data
    with DecompressionStream(io.BytesIO(blob), 1 << 20) as ds:
        assert ds.ReadToEnd(1 << 22) == data
    head = blob[:200000]
    used, out = 0, bytearray()
    while used + 70000 < len(head):
        fs = oracle.lib().zso_findFrameCompressedSize(head[used:], len(head) - used)
        out += oracle.decompress(head[used:used + fs], 65536); used += fs
    assert bytes(out) == data[:len(out)]
