"""zstdsharp_amd — MI355X (gfx950) zstd block compressor/decompressor behind ZstdSharp's one-shot API.

Host-side mirror of the reference's safe API (src/ZstdSharp/Compressor.cs, Decompressor.cs, ZstdException.cs,
ThrowHelper.cs): same member names in snake_case next to the original PascalCase, same argument meaning and
error behaviour, over the C ABI in include/zstd_mi355x.h.
"""
from .errors import ZstdException, ZSTD_ErrorCode
from .compressor import Compressor
from .decompressor import Decompressor
from . import _ffi
from .streams import CompressionStream, DecompressionStream, EndOfStreamException

__all__ = ["Compressor", "Decompressor", "CompressionStream", "DecompressionStream", "EndOfStreamException", "ZstdException", "ZSTD_ErrorCode", "_ffi"]
