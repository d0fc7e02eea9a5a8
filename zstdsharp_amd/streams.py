"""Mirrors of ZstdSharp.CompressionStream / DecompressionStream (S/CompressionStream.cs, S/DecompressionStream.cs) on
top of the library's ZSTD_compressStream2 / ZSTD_decompressStream, which batch the data through the GPU engine.
Same control flow as the reference's Write/Flush/Dispose and Read loops, so the tests read like T/ZstdNetSteamingTests.cs.
"""
import ctypes

from . import _ffi
from .compressor import Compressor
from .decompressor import Decompressor
from .errors import ensure_zstd_success

ZSTD_e_continue, ZSTD_e_flush, ZSTD_e_end = 0, 1, 2


class ZSTD_inBuffer(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("size", ctypes.c_size_t), ("pos", ctypes.c_size_t)]


class ZSTD_outBuffer(ctypes.Structure):
    _fields_ = [("dst", ctypes.c_void_p), ("size", ctypes.c_size_t), ("pos", ctypes.c_size_t)]


class EndOfStreamException(EOFError):
    pass


class CompressionStream:
    """S/CompressionStream.cs: Write() feeds ZSTD_e_continue until the input is consumed; Flush() and Dispose() run
    ZSTD_e_end until nothing remains (S/CompressionStream.cs:113-147)."""

    def __init__(self, stream, level: int = 0, bufferSize: int = 0, compressor: Compressor = None, leaveOpen: bool = True):
        self._lib = _ffi.load()
        self.innerStream = stream
        self._own = compressor is None
        self.compressor = compressor if compressor is not None else Compressor(level)
        self._outSize = bufferSize if bufferSize > 0 else ensure_zstd_success(self._lib, self._lib.ZSTD_CStreamOutSize())      # S/CompressionStream.cs:40-41
        self._out = ctypes.create_string_buffer(self._outSize)
        self._leaveOpen = leaveOpen
        self._disposed = False

    def SetParameter(self, parameter, value):
        self.compressor.SetParameter(parameter, value)

    def LoadDictionary(self, dict_bytes):                     # S/CompressionStream.cs:58-62
        self.compressor.LoadDictionary(dict_bytes)

    def _write_internal(self, data, last: bool):
        if self._disposed:
            raise ValueError("ObjectDisposedException: CompressionStream")
        n = len(data) if data is not None else 0
        keep = (ctypes.c_char * n).from_buffer_copy(data) if n else None
        inp = ZSTD_inBuffer(ctypes.addressof(keep) if n else None, n, 0)
        while True:
            out = ZSTD_outBuffer(ctypes.addressof(self._out), self._outSize, 0)
            remaining = ensure_zstd_success(self._lib, self._lib.ZSTD_compressStream2(
                self.compressor.cctx, ctypes.byref(out), ctypes.byref(inp), ZSTD_e_end if last else ZSTD_e_continue))
            if out.pos:
                self.innerStream.write(self._out.raw[:out.pos])
            if (remaining == 0) if last else (inp.pos >= inp.size):
                break

    def Write(self, buffer, offset: int = 0, count: int = None):
        mv = memoryview(buffer).cast("B")
        count = len(mv) - offset if count is None else count
        self._write_internal(bytes(mv[offset:offset + count]), False)

    def Flush(self):
        self._write_internal(None, True)

    def Dispose(self):
        if self._disposed:
            return
        try:
            self._write_internal(None, True)
        finally:
            self._disposed = True
            if self._own:
                self.compressor.Dispose()
            if not self._leaveOpen:
                self.innerStream.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.Dispose()


class DecompressionStream:
    """S/DecompressionStream.cs:79-112: Read() loops ZSTD_decompressStream, refilling its input buffer from the inner
    stream; end of input with an unfinished frame raises EndOfStreamException("Premature end of stream")."""

    def __init__(self, stream, bufferSize: int = 0, decompressor: Decompressor = None, leaveOpen: bool = True):
        self._lib = _ffi.load()
        self.innerStream = stream
        self._own = decompressor is None
        self.decompressor = decompressor if decompressor is not None else Decompressor()
        self._inSize = bufferSize if bufferSize > 0 else ensure_zstd_success(self._lib, self._lib.ZSTD_CStreamInSize())          # S/DecompressionStream.cs:41 (the reference asks the C-stream size here)
        self._inBuf = ctypes.create_string_buffer(self._inSize)
        self._input = ZSTD_inBuffer(ctypes.addressof(self._inBuf), 0, 0)
        self._last = 0
        self._leaveOpen = leaveOpen
        self._disposed = False

    def SetParameter(self, parameter, value):                 # S/DecompressionStream.cs:48-52
        self.decompressor.SetParameter(parameter, value)

    def LoadDictionary(self, dict_bytes):                     # S/DecompressionStream.cs:58-62
        self.decompressor.LoadDictionary(dict_bytes)

    def Read(self, count: int) -> bytes:
        if self._disposed:
            raise ValueError("ObjectDisposedException: DecompressionStream")
        dst = ctypes.create_string_buffer(max(count, 1))
        out = ZSTD_outBuffer(ctypes.addressof(dst), count, 0)
        while out.pos < out.size:
            if self._input.pos >= self._input.size:
                chunk = self.innerStream.read(self._inSize)
                if not chunk:
                    if self._last != 0:
                        raise EndOfStreamException("Premature end of stream")
                    break
                ctypes.memmove(self._inBuf, chunk, len(chunk))
                self._input.size, self._input.pos = len(chunk), 0
            self._last = ensure_zstd_success(self._lib, self._lib.ZSTD_decompressStream(
                self.decompressor.dctx, ctypes.byref(out), ctypes.byref(self._input)))
        return dst.raw[:out.pos]

    def ReadToEnd(self, piece: int = 1 << 16) -> bytes:
        parts = []
        while True:
            b = self.Read(piece)
            if not b:
                return b"".join(parts)
            parts.append(b)

    def Dispose(self):
        if self._disposed:
            return
        self._disposed = True
        if self._own:
            self.decompressor.Dispose()
        if not self._leaveOpen:
            self.innerStream.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.Dispose()
