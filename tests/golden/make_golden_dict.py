"""Generates the second set of golden fixtures (tests/golden/dict_*.zst, stream_*.zst, *.dict + manifest_dict.json): frames whose
bytes were NOT made by this repo's oracle — a dictionary trained by libzstd's ZDICT_trainFromBuffer and frames compressed against it
(formatted and raw-content), frames from ZSTD_compressStream2 without a pledged size (no content size in the header), and a
windowLog-11 + checksum stream (T/ZstdNetSteamingTests.cs:269-318's parameter corner).  Run ONCE in the authoring container (needs
the third-party libzstd 1.5.7 shared object bundled with Pillow there); outputs are committed, tests only read them.

Same reasoning as make_golden.py: the reference (C#) cannot run here and ships no vectors; its own T/ZstdTest.cs:69-90 treats
native libzstd as its byte-for-byte equal, so libzstd's frames stand in for "frames the reference emits" when pinning the oracle
decoder and the GPU decoder on dictionaries and unsized frames (rows f-3, f-4)."""
import ctypes, glob, hashlib, json, os, sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import datagen

sz, vp, ci = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int


class InBuf(ctypes.Structure):
    _fields_ = [("src", vp), ("size", sz), ("pos", sz)]


class OutBuf(ctypes.Structure):
    _fields_ = [("dst", vp), ("size", sz), ("pos", sz)]


def load():
    path = glob.glob("/usr/local/lib/python3*/dist-packages/pillow.libs/libzstd*")[0]
    l = ctypes.CDLL(path)
    l.ZSTD_versionNumber.restype = ctypes.c_uint
    l.ZSTD_compressBound.restype = sz; l.ZSTD_compressBound.argtypes = [sz]
    l.ZSTD_createCCtx.restype = vp
    l.ZSTD_freeCCtx.argtypes = [vp]
    l.ZSTD_CCtx_setParameter.restype = sz; l.ZSTD_CCtx_setParameter.argtypes = [vp, ci, ci]
    l.ZSTD_compress_usingDict.restype = sz; l.ZSTD_compress_usingDict.argtypes = [vp, vp, sz, vp, sz, vp, sz, ci]
    l.ZSTD_compressStream2.restype = sz; l.ZSTD_compressStream2.argtypes = [vp, ctypes.POINTER(OutBuf), ctypes.POINTER(InBuf), ci]
    l.ZDICT_trainFromBuffer.restype = sz; l.ZDICT_trainFromBuffer.argtypes = [vp, sz, vp, ctypes.POINTER(sz), ctypes.c_uint]
    l.ZDICT_isError.restype = ctypes.c_uint; l.ZDICT_isError.argtypes = [sz]
    l.ZSTD_isError.restype = ctypes.c_uint; l.ZSTD_isError.argtypes = [sz]
    return l


def words(n, seed):
    """short records over a shared vocabulary: the kind of data dictionaries are for"""
    return datagen.gen("text", n, seed)


def stream_compress(l, data, level, checksum, window_log, piece):
    """ZSTD_compressStream2 fed in pieces, ended with ZSTD_e_end, NO pledged source size -> the header carries no content size"""
    c = l.ZSTD_createCCtx()
    assert not l.ZSTD_isError(l.ZSTD_CCtx_setParameter(c, 100, level))
    assert not l.ZSTD_isError(l.ZSTD_CCtx_setParameter(c, 201, checksum))
    if window_log:
        assert not l.ZSTD_isError(l.ZSTD_CCtx_setParameter(c, 101, window_log))
    cap = l.ZSTD_compressBound(len(data)) + 1024
    dst = ctypes.create_string_buffer(cap); out = OutBuf(ctypes.addressof(dst), cap, 0)
    src = ctypes.create_string_buffer(data, len(data))
    pos = 0
    while pos < len(data):
        n = min(piece, len(data) - pos)
        inp = InBuf(ctypes.addressof(src) + pos, n, 0)
        while inp.pos < inp.size:
            r = l.ZSTD_compressStream2(c, ctypes.byref(out), ctypes.byref(inp), 0); assert not l.ZSTD_isError(r)
        pos += n
    inp = InBuf(None, 0, 0)
    while True:
        r = l.ZSTD_compressStream2(c, ctypes.byref(out), ctypes.byref(inp), 2); assert not l.ZSTD_isError(r)
        if r == 0: break
    l.ZSTD_freeCCtx(c)
    return dst.raw[:out.pos]


def main():
    l = load()
    assert l.ZSTD_versionNumber() == 10507
    cases = []
    # ---- a trained dictionary (ZDICT_trainFromBuffer = what S/DictBuilder.cs:TrainFromBuffer calls) ----
    samples = [words(300 + 37 * (i % 23), 1000 + i) for i in range(3000)]
    flat = b"".join(samples); sizes = (sz * len(samples))(*[len(s) for s in samples])
    cap = 16384; buf = ctypes.create_string_buffer(cap)
    n = l.ZDICT_trainFromBuffer(buf, cap, flat, sizes, len(samples)); assert not l.ZDICT_isError(n), n
    trained = buf.raw[:n]
    assert trained[:4] == bytes([0x37, 0xA4, 0x30, 0xEC])
    open(os.path.join(HERE, "trained_16k.dict"), "wb").write(trained)
    raw = words(6000, 4242)                                   # raw content: no magic -> history only, no dictID
    open(os.path.join(HERE, "rawcontent_6000.dict"), "wb").write(raw)

    def add(name, blob, data, **kw):
        open(os.path.join(HERE, name + ".zst"), "wb").write(blob)
        cases.append(dict(file=name + ".zst", n=len(data), sha256=hashlib.sha256(data).hexdigest(), csize=len(blob), libzstd=10507, **kw))

    def with_dict(name, data, dic, dic_file, level):
        c = l.ZSTD_createCCtx(); cap = l.ZSTD_compressBound(len(data)) + 64; dst = ctypes.create_string_buffer(cap)
        r = l.ZSTD_compress_usingDict(c, dst, cap, data, len(data), dic, len(dic), level); assert not l.ZSTD_isError(r)
        l.ZSTD_freeCCtx(c)
        add(name, dst.raw[:r], data, dict=dic_file, level=level, kind="text")

    for i, (n_, level) in enumerate(((120, 1), (900, 1), (5000, 1), (40000, 1), (900, 5), (40000, 5), (200000, 3))):
        with_dict(f"dict_fmt_{n_}_l{level}", words(n_, 7000 + i), trained, "trained_16k.dict", level)
    for i, (n_, level) in enumerate(((500, 1), (30000, 1), (30000, 5), (150000, 3))):
        with_dict(f"dict_raw_{n_}_l{level}", words(n_, 7100 + i), raw, "rawcontent_6000.dict", level)
    # ---- frames without a content size: the streaming compressor with nothing pledged ----
    for kind, n_, level, chk, piece in (("text", 300000, 1, 0, 50000), ("mixed", 500000, 5, 1, 131072), ("zipf", 70000, 3, 0, 1000), ("bytei", 1024, 1, 1, 1)):
        data = datagen.gen(kind, n_, 8000 + n_)
        add(f"stream_unsized_{kind}_{n_}_l{level}", stream_compress(l, data, level, chk, 0, piece), data, level=level, kind=kind, checksum=chk, unsized=1)
    # ---- windowLog 11 + checksum (T/ZstdNetSteamingTests.cs:293: SetParameter(windowLog, 11) + checksumFlag) ----
    data = datagen.gen("text", 1 << 20, 8111)
    add("stream_w11_chk_text_1m_l1", stream_compress(l, data, 1, 1, 11, 4096), data, level=1, kind="text", checksum=1, unsized=1, windowLog=11)
    json.dump(dict(generator="tests/golden/make_golden_dict.py", cases=cases), open(os.path.join(HERE, "manifest_dict.json"), "w"), indent=1)
    print(len(cases), "fixtures,", sum(c["csize"] for c in cases), "bytes; dictionary", len(trained), "bytes")


if __name__ == "__main__":
    main()
