"""Generates tests/golden/*.zst + manifest.json.  Run ONCE in the authoring container (it needs third-party libzstd
shared objects that exist there: 1.4.8 system, 1.5.7 bundled with Pillow); the outputs are committed, the GPU box and
the test-suite only read them.

Why libzstd: the reference (ZstdSharp, C#) cannot be built or run here and ships no compressed vectors; its own
test T/ZstdTest.cs:69-90 asserts byte-identity between ZstdSharp and native libzstd at every level, i.e. libzstd is
the codec the reference accepts as its equal.  Frames below therefore stand in for "frames the reference would
emit" when pinning the oracle decoder and the GPU decoder; levels/sizes follow the reference's fixtures
(GenerateBuffer sizes 2..99002 step 3000, T/ZstdNetTests.cs:478-496; 1 KiB / 1 MiB i%256, T/ZstdNetSteamingTests.cs:22-43).
Inputs are regenerated from tests/datagen.py (seeded), only their sha256 is stored.
"""
import ctypes, glob, hashlib, json, os, sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import datagen


def load(path):
    l = ctypes.CDLL(path)
    l.ZSTD_compress.restype = ctypes.c_size_t
    l.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    l.ZSTD_compressBound.restype = ctypes.c_size_t
    l.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    l.ZSTD_versionNumber.restype = ctypes.c_uint
    l.ZSTD_createCCtx.restype = ctypes.c_void_p
    l.ZSTD_CCtx_setParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    l.ZSTD_compress2.restype = ctypes.c_size_t
    l.ZSTD_compress2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    return l


def compress(l, data, level, checksum=0):
    cap = l.ZSTD_compressBound(len(data)) + 16
    buf = ctypes.create_string_buffer(cap)
    c = l.ZSTD_createCCtx()
    l.ZSTD_CCtx_setParameter(c, 100, level); l.ZSTD_CCtx_setParameter(c, 201, checksum)
    n = l.ZSTD_compress2(c, buf, cap, data, len(data))
    l.ZSTD_freeCCtx(ctypes.c_void_p(c))
    return buf.raw[:n]


def main():
    libs = {}
    for p in ["/usr/lib/x86_64-linux-gnu/libzstd.so.1"] + glob.glob("/usr/local/lib/python3*/dist-packages/pillow.libs/libzstd*"):
        l = load(p); libs[l.ZSTD_versionNumber()] = l
    new, old = libs[max(libs)], libs[min(libs)]
    cases = []
    def add(name, kind, n, seed, level, lib, checksum=0, multi=1):
        data = datagen.gen(kind, n, seed)
        blob = b"".join(compress(lib, data, level, checksum) for _ in range(multi))
        if multi > 1:
            blob = blob[:len(blob) // multi] + b"\x50\x2a\x4d\x18\x04\x00\x00\x00skip" + blob[len(blob) // multi:]   # a skippable frame between frames
        fn = name + ".zst"
        open(os.path.join(HERE, fn), "wb").write(blob)
        cases.append(dict(file=fn, kind=kind, n=n, seed=seed, level=level, checksum=checksum, copies=multi,
                          libzstd=lib.ZSTD_versionNumber(), sha256=hashlib.sha256(data * multi).hexdigest(), csize=len(blob)))
    for n in (0, 1, 2, 3002, 12002, 99002):
        add(f"bytei_{n}_l1", "bytei", n, 0, 1, new)
    add("bytei_1024_l5", "bytei", 1024, 0, 5, new)
    add("text_20000_l1", "text", 20000, 3, 1, new)
    add("text_70000_l5", "text", 70000, 4, 5, new)
    add("text_300000_l5", "text", 300000, 5, 5, new)          # multi-block, history across blocks, repeat tables
    add("text_300000_l3_old", "text", 300000, 5, 3, old)
    add("zipf_65536_l1", "zipf", 65536, 1234, 1, new)
    add("zipf_40000_l5_chk", "zipf", 40000, 99, 5, new, checksum=1)
    add("runs_50000_l1", "runs", 50000, 11, 1, new)            # RLE blocks / RLE literals / long matches
    add("zeros_200000_l1", "zeros", 200000, 0, 1, new)
    add("rand_5000_l1", "rand", 5000, 8, 1, new)               # raw block
    add("mixed_150000_l5", "mixed", 150000, 21, 5, new)
    add("mixed_150000_l19", "mixed", 150000, 21, 19, new)      # optimal parser output: many repcodes, treeless literals
    add("period_9000_l1", "period", 9000, 13, 1, new)
    add("text_5000_x2_multiframe", "text", 5000, 17, 1, new, multi=2)
    json.dump(dict(generator="tests/golden/make_golden.py", cases=cases), open(os.path.join(HERE, "manifest.json"), "w"), indent=1)
    print(len(cases), "fixtures,", sum(c["csize"] for c in cases), "bytes")


if __name__ == "__main__":
    main()
