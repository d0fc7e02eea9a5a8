"""Mirror of ZstdSharp.Compressor (S/Compressor.cs) over libzstd_mi355x.so."""
import ctypes

from . import _ffi
from .errors import DST_SIZE_TOO_SMALL, ensure_zstd_success

ZSTD_c_compressionLevel = 100


def _as_buffer(data):
    """bytes / bytearray / memoryview / numpy -> (address, length, keepalive)"""
    if isinstance(data, bytes):
        return ctypes.cast(ctypes.c_char_p(data), ctypes.c_void_p).value if data else None, len(data), data
    mv = memoryview(data).cast("B")
    if len(mv) == 0:
        return None, 0, mv
    if mv.readonly:
        b = bytes(mv)
        return ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p).value, len(b), b
    arr = (ctypes.c_ubyte * len(mv)).from_buffer(mv)
    return ctypes.addressof(arr), len(mv), (arr, mv)


class Compressor:
    """S/Compressor.cs:7-163.  One instance is used by one thread at a time."""

    def __init__(self, level: int = 0, device: int = None):
        self._lib = _ffi.load()
        self.cctx = self._lib.ZSTD_createCCtx()          # S/Compressor.cs:32
        if not self.cctx:
            raise MemoryError("ZSTD_createCCtx")
        if device is not None:
            ensure_zstd_success(self._lib, self._lib.ZSTDMI_CCtx_setDevice(self.cctx, device))
        self._level = 0
        self.Level = level if level else self.DefaultCompressionLevel

    # ---- static members (S/Compressor.cs:8-10) ----
    @staticmethod
    def _static(name):
        return getattr(_ffi.load(), name)()

    MinCompressionLevel = property(lambda self: self._lib.ZSTD_minCLevel())
    MaxCompressionLevel = property(lambda self: self._lib.ZSTD_maxCLevel())
    DefaultCompressionLevel = 3

    # ---- Level (S/Compressor.cs:16-27) ----
    @property
    def Level(self):
        return self._level

    @Level.setter
    def Level(self, value):
        if self._level != value:
            self._level = value
            self.SetParameter(ZSTD_c_compressionLevel, value)

    level = Level

    def SetParameter(self, parameter: int, value: int):      # S/Compressor.cs:46-50
        self._ensure_not_disposed()
        ensure_zstd_success(self._lib, self._lib.ZSTD_CCtx_setParameter(self.cctx, int(parameter), int(value)))

    def GetParameter(self, parameter: int) -> int:            # S/Compressor.cs:52-57
        self._ensure_not_disposed()
        v = ctypes.c_int(0)
        ensure_zstd_success(self._lib, self._lib.ZSTD_CCtx_getParameter(self.cctx, int(parameter), ctypes.byref(v)))
        return v.value

    def LoadDictionary(self, dict_bytes):                     # S/Compressor.cs:36-43
        self._ensure_not_disposed()
        addr, n, keep = _as_buffer(dict_bytes if dict_bytes is not None else b"")
        ensure_zstd_success(self._lib, self._lib.ZSTD_CCtx_loadDictionary(self.cctx, addr, n))

    @staticmethod
    def GetCompressBound(length: int) -> int:                  # S/Compressor.cs:72-76
        return _ffi.load().ZSTD_compressBound(length)

    # ---- Wrap (S/Compressor.cs:78-96) ----
    def Wrap(self, src, dest=None, offset: int = 0):
        """Wrap(src) -> bytes;  Wrap(src, dest[, offset]) -> number of bytes written into dest."""
        self._ensure_not_disposed()
        saddr, sn, skeep = _as_buffer(src)
        if dest is None:
            cap = self.GetCompressBound(sn)
            out = ctypes.create_string_buffer(max(cap, 1))
            n = ensure_zstd_success(self._lib, self._lib.ZSTD_compress2(self.cctx, out, cap, saddr, sn))
            return out.raw[:n]
        daddr, dn, dkeep = _as_buffer(dest)
        if offset < 0 or offset > dn:
            raise ValueError("offset")
        return ensure_zstd_success(self._lib, self._lib.ZSTD_compress2(self.cctx, (daddr or 0) + offset if daddr else None, dn - offset, saddr, sn))

    def TryWrap(self, src, dest, offset: int = 0):             # S/Compressor.cs:98-122
        """-> (ok, written): ok is False exactly when dest is too small (ZSTD_error_dstSize_tooSmall)."""
        self._ensure_not_disposed()
        saddr, sn, skeep = _as_buffer(src)
        daddr, dn, dkeep = _as_buffer(dest)
        r = self._lib.ZSTD_compress2(self.cctx, (daddr + offset) if daddr else None, dn - offset, saddr, sn)
        if r == DST_SIZE_TOO_SMALL:
            return False, 0
        return True, ensure_zstd_success(self._lib, r)

    wrap, try_wrap, set_parameter, get_parameter, load_dictionary = Wrap, TryWrap, SetParameter, GetParameter, LoadDictionary

    # ---- lifetime (S/Compressor.cs:59-70, 124-147) ----
    def Dispose(self):
        if getattr(self, "cctx", None):
            self._lib.ZSTD_freeCCtx(self.cctx)
            self.cctx = None

    dispose = close = Dispose

    def _ensure_not_disposed(self):
        if not self.cctx:
            raise RuntimeError("ObjectDisposedException: Compressor")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.Dispose()

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass
