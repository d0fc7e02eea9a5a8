"""Mirror of ZstdSharp.Decompressor (S/Decompressor.cs) over libzstd_mi355x.so."""
import ctypes

from . import _ffi
from .compressor import _as_buffer
from .errors import DST_SIZE_TOO_SMALL, ZstdException, ZSTD_ErrorCode, ensure_content_size_ok, ensure_zstd_success


class Decompressor:
    """S/Decompressor.cs:7-148."""

    def __init__(self, device: int = None):
        self._lib = _ffi.load()
        self.dctx = self._lib.ZSTD_createDCtx()          # S/Decompressor.cs:12
        if not self.dctx:
            raise MemoryError("ZSTD_createDCtx")
        if device is not None:
            ensure_zstd_success(self._lib, self._lib.ZSTDMI_DCtx_setDevice(self.dctx, device))

    def SetParameter(self, parameter: int, value: int):      # S/Decompressor.cs:39-43
        self._ensure_not_disposed()
        ensure_zstd_success(self._lib, self._lib.ZSTD_DCtx_setParameter(self.dctx, int(parameter), int(value)))

    def GetParameter(self, parameter: int) -> int:            # S/Decompressor.cs:45-50
        self._ensure_not_disposed()
        v = ctypes.c_int(0)
        ensure_zstd_success(self._lib, self._lib.ZSTD_DCtx_getParameter(self.dctx, int(parameter), ctypes.byref(v)))
        return v.value

    def LoadDictionary(self, dict_bytes):                     # S/Decompressor.cs:29-36
        self._ensure_not_disposed()
        addr, n, keep = _as_buffer(dict_bytes if dict_bytes is not None else b"")
        ensure_zstd_success(self._lib, self._lib.ZSTD_DCtx_loadDictionary(self.dctx, addr, n))

    @staticmethod
    def GetDecompressedSize(src) -> int:                      # S/Decompressor.cs:50-54
        addr, n, keep = _as_buffer(src)
        return ensure_content_size_ok(_ffi.load().ZSTD_decompressBound(addr, n))

    def Unwrap(self, src, dest=None, offset: int = 0, maxDecompressedSize: int = (1 << 31) - 1):
        """Unwrap(src[, maxDecompressedSize=..]) -> bytes;  Unwrap(src, dest[, offset]) -> bytes written (S/Decompressor.cs:56-88)."""
        self._ensure_not_disposed()
        saddr, sn, skeep = _as_buffer(src)
        if dest is None:
            expected = self.GetDecompressedSize(src)
            if expected > maxDecompressedSize:
                raise ZstdException(ZSTD_ErrorCode.ZSTD_error_dstSize_tooSmall,
                                    f"Decompressed content size {expected} is greater than {maxDecompressedSize}")
            out = ctypes.create_string_buffer(max(expected, 1))
            n = ensure_zstd_success(self._lib, self._lib.ZSTD_decompressDCtx(self.dctx, out, expected, saddr, sn))
            return out.raw[:n]          # new Span<byte>(dest, 0, length): `expected` is a bound, not a promise (S/Decompressor.cs:63-75)
        daddr, dn, dkeep = _as_buffer(dest)
        return ensure_zstd_success(self._lib, self._lib.ZSTD_decompressDCtx(self.dctx, (daddr + offset) if daddr else None, dn - offset, saddr, sn))

    def TryUnwrap(self, src, dest, offset: int = 0):          # S/Decompressor.cs:90-111
        self._ensure_not_disposed()
        saddr, sn, skeep = _as_buffer(src)
        daddr, dn, dkeep = _as_buffer(dest)
        r = self._lib.ZSTD_decompressDCtx(self.dctx, (daddr + offset) if daddr else None, dn - offset, saddr, sn)
        if r == DST_SIZE_TOO_SMALL:
            return False, 0
        return True, ensure_zstd_success(self._lib, r)

    unwrap, try_unwrap, set_parameter, get_parameter, load_dictionary, get_decompressed_size = \
        Unwrap, TryUnwrap, SetParameter, GetParameter, LoadDictionary, GetDecompressedSize

    def Dispose(self):                                        # S/Decompressor.cs:113-147
        if getattr(self, "dctx", None):
            self._lib.ZSTD_freeDCtx(self.dctx)
            self.dctx = None

    dispose = close = Dispose

    def _ensure_not_disposed(self):
        if not self.dctx:
            raise RuntimeError("ObjectDisposedException: Decompressor")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.Dispose()

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass
