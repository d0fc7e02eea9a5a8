"""CPU tests: the C-ABI library loads, exports every symbol its header declares, and the no-compute entry points
behave as the reference's (compressBound, levels, error names, parameters).  No kernel is launched here."""
import ctypes
import os
import re

import pytest

import zstdsharp_amd
from zstdsharp_amd import _ffi
from zstdsharp_amd.errors import ZSTD_ErrorCode, ZstdException, is_error, get_error_code

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "zstd_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:ZSTD(?:MI)?|ZDICT)_[A-Za-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 44 and "ZDICT_isError" in names and "ZSTD_CStreamOutSize" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/zstd_mi355x.h but not exported"
        assert n in _ffi.SIGNATURES, f"{n} has no ctypes signature"


def test_compress_bound_matches_reference_formula():
    """U/ZstdCompress.cs:19-22."""
    lib = _ffi.load()
    for n in (0, 1, 255, 65536, 131071, 131072, 1 << 20, 10192446):
        want = n + (n >> 8) + (((128 << 10) - n) >> 11 if n < (128 << 10) else 0)
        assert lib.ZSTD_compressBound(n) == want
    assert lib.ZSTD_compressBound(65536) == 65824


def test_levels_and_version():
    lib = _ffi.load()
    assert lib.ZSTD_minCLevel() == -131072 and lib.ZSTD_maxCLevel() == 22 and lib.ZSTD_defaultCLevel() == 3
    assert lib.ZSTD_versionNumber() == 10501 and lib.ZSTD_versionString() == b"1.5.1"


def test_stream_buffer_sizes_and_zdict_helpers():
    """What S/CompressionStream.cs:41, S/DecompressionStream.cs:41 and S/ThrowHelper.cs:18-24 bind to
    (U/ZstdCompress.cs:6241-6249, U/ZstdDecompress.cs:2096-2104, U/Zdict.cs:11-19)."""
    lib = _ffi.load()
    assert lib.ZSTD_CStreamInSize() == 1 << 17
    assert lib.ZSTD_CStreamOutSize() == lib.ZSTD_compressBound(1 << 17) + 3 + 4 == 131591
    assert lib.ZSTD_DStreamInSize() == (1 << 17) + 3 and lib.ZSTD_DStreamOutSize() == 1 << 17
    assert lib.ZDICT_isError((1 << 64) - 34) == 1 and lib.ZDICT_isError(1000) == 0
    assert lib.ZDICT_getErrorName((1 << 64) - 34) == b"Cannot create Dictionary from provided samples"


def test_error_convention():
    lib = _ffi.load()
    for code, text in ((70, b"Destination buffer is too small"), (20, b"Corrupted block detected"), (10, b"Unknown frame descriptor"),
                       (72, b"Src size is incorrect"), (22, b"Restored data doesn't match checksum"), (40, b"Unsupported parameter")):
        v = (1 << 64) - code
        assert lib.ZSTD_isError(v) == 1 and is_error(v) and get_error_code(v) == code
        assert lib.ZSTD_getErrorName(v) == text
    assert lib.ZSTD_isError(12345) == 0 and lib.ZSTD_getErrorName(0) == b"No error detected"


def test_parameters_without_a_device():
    """setParameter/getParameter are host-side state (S/Compressor.cs:46-57); level 0 means 3 (U/ZstdCompress.cs:896-899)."""
    c = zstdsharp_amd.Compressor(1)
    assert c.Level == 1 and c.GetParameter(100) == 1
    c.Level = 0
    assert c.GetParameter(100) == 3
    c.SetParameter(201, 1)
    assert c.GetParameter(201) == 1
    c.SetParameter(100, 99)
    assert c.GetParameter(100) == 22                       # clamped to ZSTD_maxCLevel
    with pytest.raises(ZstdException) as e:
        c.SetParameter(400, 2)                             # nbWorkers: unsupported, as in the reference (U/ZstdCompress.cs:1064-1072)
    assert e.value.Code == ZSTD_ErrorCode.ZSTD_error_parameter_unsupported
    # match-finder parameters: bounds as ZSTD_cParam_getBounds; values the kernels implement are accepted and
    # read back, other in-range values are refused loudly, 0 restores "from the level"
    c.Level = 1
    for param, ok, unsupported, out_of_bound in ((107, (1, 2, 3, 5, 9), (), (10, -1)),           # strategy
                                                 (106, (0, 1, 7, 131072), (), (131073, -1)),      # targetLength
                                                 (102, (13, 0), (12, 17), (5, 31)),               # hashLog
                                                 (105, (6, 0), (5, 4), (2, 8)),                   # minMatch (fast strategy: 6)
                                                 (103, (12, 0), (13,), (5, 31)),                  # chainLog (level 1, <= 128 KiB: 12)
                                                 (104, (1, 2, 5, 30, 0), (), (31,))):              # searchLog (attempts of the level >= 5 search)
        for v in ok:
            c.SetParameter(param, v)
            assert c.GetParameter(param) == v
        for v in unsupported:
            with pytest.raises(ZstdException) as e:
                c.SetParameter(param, v)
            assert e.value.Code == ZSTD_ErrorCode.ZSTD_error_parameter_unsupported, (param, v)
        for v in out_of_bound:
            with pytest.raises(ZstdException) as e:
                c.SetParameter(param, v)
            assert e.value.Code == ZSTD_ErrorCode.ZSTD_error_parameter_outOfBound, (param, v)
        c.SetParameter(param, 0)
    # frame parameters: ZSTD_c_windowLog 10 .. 31 (below 16: frames of 1 << windowLog bytes), ZSTD_c_contentSizeFlag 0 / 1
    for v in (10, 11, 15, 16, 17, 27, 31, 0):
        c.SetParameter(101, v); assert c.GetParameter(101) == v
    for v in (9, 32, -1):
        with pytest.raises(ZstdException) as e:
            c.SetParameter(101, v)
        assert e.value.Code == ZSTD_ErrorCode.ZSTD_error_parameter_outOfBound
    c.SetParameter(200, 0); assert c.GetParameter(200) == 0
    c.SetParameter(200, 1); assert c.GetParameter(200) == 1
    c.SetParameter(107, 2)
    c.SetParameter(105, 5)                                 # doubleFast's short hash is 5 bytes wide
    c.SetParameter(105, 0); c.SetParameter(107, 0)
    c.Level = -5
    assert c.GetParameter(100) == -5                       # negative levels are kept (U/ZstdCompress.cs:886-905)
    c.Level = 1
    c.LoadDictionary(b"some dictionary bytes")             # raw content: kept on the host until a compression needs it
    c.LoadDictionary(None)
    c.LoadDictionary(bytes([0x37, 0xA4, 0x30, 0xEC]) + bytes(64))          # formatted: its header is validated on the device, i.e. at
    with pytest.raises(ZstdException) as e:                                # first use when none is there yet — and there is none here
        c.Wrap(b"x" * 100)
    assert e.value.Code == ZSTD_ErrorCode.ZSTD_error_init_missing
    c.LoadDictionary(None)
    c.Dispose()
    with pytest.raises(RuntimeError):
        c.Wrap(b"x")
    d = zstdsharp_amd.Decompressor()
    d.SetParameter(100, 25)
    assert d.GetParameter(100) == 25
    d.Dispose()


def test_free_null_contexts():
    lib = _ffi.load()
    assert lib.ZSTD_freeCCtx(None) == 0 and lib.ZSTD_freeDCtx(None) == 0


def test_decompress_bound_is_a_header_walk(oracle, golden):
    """ZSTD_decompressBound never touches a kernel: it must agree with the oracle on every golden frame."""
    lib = _ffi.load()
    for c in golden:
        blob = open(c["path"], "rb").read()
        assert lib.ZSTD_decompressBound(blob, len(blob)) == c["n"] * c["copies"]
        assert lib.ZSTD_findFrameCompressedSize(blob, len(blob)) == oracle.lib().zso_findFrameCompressedSize(blob, len(blob))
    assert lib.ZSTD_decompressBound(b"garbage!!garbage!!", 18) == (1 << 64) - 2
    assert zstdsharp_amd.Decompressor.GetDecompressedSize(open(golden[3]["path"], "rb").read()) == golden[3]["n"]


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a gfx950 device a compute call must raise, never return data."""
    lib = _ffi.load()
    if lib.ZSTDMI_deviceCount() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(ZstdException) as e:
        zstdsharp_amd.Compressor(1).Wrap(b"hello world")
    assert e.value.Code == ZSTD_ErrorCode.ZSTD_error_init_missing
    with pytest.raises(ZstdException):
        zstdsharp_amd.Decompressor().Unwrap(bytes([0x28, 0xB5, 0x2F, 0xFD, 0x20, 0x00, 0x01, 0x00, 0x00]))


def test_product_never_links_or_imports_the_oracle():
    """The oracle is test infrastructure: nothing in the shipped package may link, load or call it."""
    import subprocess
    out = subprocess.run(["ldd", _ffi.LIB_PATH], capture_output=True, text=True).stdout
    assert "zso" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "zstdsharp_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for token in ("zso_", "libzso", "oracle_lib", "import oracle", "oracle/"):
                    assert token not in text, (f, token)
