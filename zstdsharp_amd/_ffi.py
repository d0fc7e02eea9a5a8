"""ctypes binding of libzstd_mi355x.so (the C ABI declared in include/zstd_mi355x.h).

The library is the product; there is no Python or CPU codec behind it.  If the shared object is missing or has
no gfx950 device to run on, calls fail loudly (ZstdException / OSError) — they never fall back.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzstd_mi355x.so")

c_size_t, c_void_p, c_int, c_uint, c_ull = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_ulonglong


class ZSTDMI_Seq(ctypes.Structure):
    _fields_ = [("offBase", ctypes.c_uint32), ("litLength", ctypes.c_uint16), ("mlBase", ctypes.c_uint16)]


# name -> (restype, argtypes); every symbol include/zstd_mi355x.h declares
SIGNATURES = {
    "ZSTD_createCCtx": (c_void_p, []),
    "ZSTD_freeCCtx": (c_size_t, [c_void_p]),
    "ZSTD_CCtx_setParameter": (c_size_t, [c_void_p, c_int, c_int]),
    "ZSTD_CCtx_getParameter": (c_size_t, [c_void_p, c_int, ctypes.POINTER(c_int)]),
    "ZSTD_CCtx_loadDictionary": (c_size_t, [c_void_p, c_void_p, c_size_t]),
    "ZSTD_compressBound": (c_size_t, [c_size_t]),
    "ZSTD_compress2": (c_size_t, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t]),
    "ZSTD_compressCCtx": (c_size_t, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_int]),
    "ZSTD_minCLevel": (c_int, []),
    "ZSTD_maxCLevel": (c_int, []),
    "ZSTD_defaultCLevel": (c_int, []),
    "ZSTD_createDCtx": (c_void_p, []),
    "ZSTD_freeDCtx": (c_size_t, [c_void_p]),
    "ZSTD_DCtx_setParameter": (c_size_t, [c_void_p, c_int, c_int]),
    "ZSTD_DCtx_getParameter": (c_size_t, [c_void_p, c_int, ctypes.POINTER(c_int)]),
    "ZSTD_DCtx_loadDictionary": (c_size_t, [c_void_p, c_void_p, c_size_t]),
    "ZSTD_decompressBound": (c_ull, [c_void_p, c_size_t]),
    "ZSTD_getFrameContentSize": (c_ull, [c_void_p, c_size_t]),
    "ZSTD_findFrameCompressedSize": (c_size_t, [c_void_p, c_size_t]),
    "ZSTD_decompressDCtx": (c_size_t, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t]),
    "ZSTD_isError": (c_uint, [c_size_t]),
    "ZSTD_getErrorName": (ctypes.c_char_p, [c_size_t]),
    "ZSTD_versionNumber": (c_uint, []),
    "ZSTD_versionString": (ctypes.c_char_p, []),
    "ZSTD_compressStream2": (c_size_t, [c_void_p, c_void_p, c_void_p, c_int]),
    "ZSTD_decompressStream": (c_size_t, [c_void_p, c_void_p, c_void_p]),
    "ZSTD_CStreamInSize": (c_size_t, []),
    "ZSTD_CStreamOutSize": (c_size_t, []),
    "ZSTD_DStreamInSize": (c_size_t, []),
    "ZSTD_DStreamOutSize": (c_size_t, []),
    "ZDICT_isError": (c_uint, [c_size_t]),
    "ZDICT_getErrorName": (ctypes.c_char_p, [c_size_t]),
    "ZSTDMI_deviceCount": (c_int, []),
    "ZSTDMI_CCtx_setDevice": (c_size_t, [c_void_p, c_int]),
    "ZSTDMI_DCtx_setDevice": (c_size_t, [c_void_p, c_int]),
    "ZSTDMI_CCtx_setDevices": (c_size_t, [c_void_p, ctypes.POINTER(c_int), c_int]),
    "ZSTDMI_DCtx_setDevices": (c_size_t, [c_void_p, ctypes.POINTER(c_int), c_int]),
    "ZSTDMI_CCtx_setStream": (c_size_t, [c_void_p, c_void_p]),
    "ZSTDMI_DCtx_setStream": (c_size_t, [c_void_p, c_void_p]),
    "ZSTDMI_CCtx_setPassChunks": (c_size_t, [c_void_p, c_uint]),
    "ZSTDMI_CCtx_setHistory": (c_size_t, [c_void_p, c_int, c_uint]),
    "ZSTDMI_DCtx_setLiteralDecoder": (c_size_t, [c_void_p, c_uint]),
    "ZSTDMI_DCtx_setLongFrames": (c_size_t, [c_void_p, c_uint]),
    "ZSTDMI_DCtx_setOverlap": (c_size_t, [c_void_p, c_uint]),
    "ZSTDMI_DCtx_setExecWaves": (c_size_t, [c_void_p, c_uint]),
    "ZSTDMI_debugLastWalkSerial": (c_int, [c_void_p]),
    "ZSTDMI_CCtx_setParser": (c_size_t, [c_void_p, c_uint]),
    "ZSTDMI_compressDevice": (c_size_t, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t]),
    "ZSTDMI_decompressDevice": (c_size_t, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t]),
    "ZSTDMI_CCtx_setProfiling": (c_size_t, [c_void_p, c_int]),
    "ZSTDMI_DCtx_setProfiling": (c_size_t, [c_void_p, c_int]),
    "ZSTDMI_CCtx_getStageTimes": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_char_p), c_int]),
    "ZSTDMI_DCtx_getStageTimes": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_char_p), c_int]),
    "ZSTDMI_debugGetChunk": (c_size_t, [c_void_p, c_size_t, ctypes.POINTER(ZSTDMI_Seq), c_size_t, ctypes.POINTER(c_size_t),
                                        c_void_p, c_size_t, ctypes.POINTER(c_size_t)]),
    "ZSTDMI_debugPoisonedChunk": (c_size_t, [c_void_p, c_uint, c_uint, c_uint, c_uint]),
    "ZSTDMI_debugEntropyBlock": (c_size_t, [c_void_p, c_void_p, c_size_t, ctypes.POINTER(ZSTDMI_Seq), c_size_t,
                                            c_void_p, c_size_t, c_size_t]),
}

_lib = None


def load():
    """Load the shared object and type every exported symbol.  Raises OSError if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"or `make -C zstdsharp_amd/csrc` (there is no fallback codec)")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError here = ABI drift between header and library
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib
