"""ctypes binding of the CPU oracle (oracle/libzso.so).  Test infrastructure only — never imported by the product."""
import ctypes
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_DIR, "libzso.so")
SIZE_MAX = (1 << 64) - 1


class ZsoSeq(ctypes.Structure):
    _fields_ = [("offBase", ctypes.c_uint32), ("litLength", ctypes.c_uint16), ("mlBase", ctypes.c_uint16)]


def build(force: bool = False):
    srcs = [os.path.join(_DIR, f) for f in ("zso_dec.c", "zso_enc.c", "zso_xxh64.c", "zso_common.h", "zso_enc.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _DIR, "libzso.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        o = ctypes.CDLL(build())
        sz, vp, ci = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int
        o.zso_decompress.restype, o.zso_decompress.argtypes = sz, [vp, sz, vp, sz]
        o.zso_decompress_usingDict.restype, o.zso_decompress_usingDict.argtypes = sz, [vp, sz, vp, sz, vp, sz]
        o.zso_compress_usingDict.restype, o.zso_compress_usingDict.argtypes = sz, [vp, sz, vp, sz, vp, sz, ci, ci]
        o.zso_make_dictionary.restype, o.zso_make_dictionary.argtypes = sz, [vp, sz, vp, sz, vp, sz, ctypes.c_uint32]
        o.zso_decompressBound.restype, o.zso_decompressBound.argtypes = ctypes.c_uint64, [vp, sz]
        o.zso_findFrameCompressedSize.restype, o.zso_findFrameCompressedSize.argtypes = sz, [vp, sz]
        o.zso_compress.restype, o.zso_compress.argtypes = sz, [vp, sz, vp, sz, ci, ci]
        o.zso_compress_chunked.restype, o.zso_compress_chunked.argtypes = sz, [vp, sz, vp, sz, ci, ci, sz]
        o.zso_compressBound.restype, o.zso_compressBound.argtypes = sz, [sz]
        o.zso_xxh64.restype, o.zso_xxh64.argtypes = ctypes.c_uint64, [vp, sz, ctypes.c_uint64]
        o.zso_block_sequences.restype = sz
        o.zso_block_sequences.argtypes = [ctypes.POINTER(ZsoSeq), sz, vp, ctypes.POINTER(sz), vp, sz, ci]
        o.zso_entropy_block.restype = sz
        o.zso_entropy_block.argtypes = [vp, sz, ctypes.POINTER(ZsoSeq), sz, vp, sz, ctypes.c_uint32, sz]
        o.zso_huf_buildLengths.restype = sz
        o.zso_huf_buildLengths.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32, ctypes.c_uint32]
        _lib = o
    return _lib


def is_error(v: int) -> bool:
    return v > SIZE_MAX - 120


def err_code(v: int) -> int:
    return (1 << 64) - v if is_error(v) else 0


def decompress(data: bytes, cap: int, dict_bytes: bytes = None):
    """-> bytes, or a negative error code.  dict_bytes: a raw-content dictionary"""
    buf = ctypes.create_string_buffer(max(cap, 1))
    if dict_bytes:
        n = lib().zso_decompress_usingDict(buf, cap, data, len(data), dict_bytes, len(dict_bytes))
    else:
        n = lib().zso_decompress(buf, cap, data, len(data))
    return -err_code(n) if is_error(n) else buf.raw[:n]


def make_dictionary(content: bytes, sample: bytes, dict_id: int) -> bytes:
    """a FORMATTED dictionary (magic, dictID, entropy tables from the sample's statistics, repcodes, content): tests only"""
    cap = len(content) + 2048
    buf = ctypes.create_string_buffer(cap)
    n = lib().zso_make_dictionary(buf, cap, content, len(content), sample, len(sample), dict_id)
    assert not is_error(n), err_code(n)
    return buf.raw[:n]


def compress_dict(data: bytes, dict_bytes: bytes, level: int = 1, checksum: int = 0):
    """one frame whose matches may reach into the raw-content dictionary (fast strategy only)"""
    o = lib()
    cap = o.zso_compressBound(len(data)) + 64
    buf = ctypes.create_string_buffer(cap)
    n = o.zso_compress_usingDict(buf, cap, data, len(data), dict_bytes, len(dict_bytes), level, checksum)
    return -err_code(n) if is_error(n) else buf.raw[:n]


def compress(data: bytes, level: int = 1, checksum: int = 0, chunk: int = 0):
    o = lib()
    cap = o.zso_compressBound(len(data)) + 64 + (len(data) // max(chunk, 1) + 1) * 32 * (1 if chunk else 0)
    buf = ctypes.create_string_buffer(cap)
    n = (o.zso_compress_chunked(buf, cap, data, len(data), level, checksum, chunk) if chunk
         else o.zso_compress(buf, cap, data, len(data), level, checksum))
    return -err_code(n) if is_error(n) else buf.raw[:n]


def entropy_block(seqs, lits: bytes, src_size: int, strategy: int = 1):
    """seqs: list of (offBase, litLength, mlBase).  -> compressed block body bytes, b'' if 'store raw', or negative error."""
    n = len(seqs)
    arr = (ZsoSeq * max(n, 1))()
    for i, (o_, l_, m_) in enumerate(seqs):
        arr[i].offBase, arr[i].litLength, arr[i].mlBase = o_, l_, m_
    cap = src_size + 1024
    buf = ctypes.create_string_buffer(cap)
    r = lib().zso_entropy_block(buf, cap, arr, n, lits, len(lits), strategy, src_size)
    return -err_code(r) if is_error(r) else buf.raw[:r]


def block_sequences(data: bytes, level: int = 1):
    """-> (list of (offBase, litLength, mlBase), literal bytes) of the reference match finder on one block."""
    cap = len(data) // 3 + 8
    arr = (ZsoSeq * cap)()
    lits = ctypes.create_string_buffer(len(data) + 64)
    ls = ctypes.c_size_t(0)
    n = lib().zso_block_sequences(arr, cap, lits, ctypes.byref(ls), data, len(data), level)
    return [(arr[i].offBase, arr[i].litLength, arr[i].mlBase) for i in range(n)], lits.raw[:ls.value]
