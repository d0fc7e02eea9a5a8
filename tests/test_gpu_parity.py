"""GPU tests (run with -m gpu on an MI355X): the HIP path through the C ABI against the CPU oracle.

Parity definition (DESIGN.md "Parity"):
  * decode     : bit-exact with the oracle decoder on every golden frame and on oracle-built frames;
  * entropy    : given the same seqStore, the GPU literals+sequences sections are BYTE-IDENTICAL to the oracle's
                 restatement of ZSTD_entropyCompressSeqStore (rows a-7..a-11);
  * match find : the GPU parse is a different (wavefront-parallel) parse than ZSTD_fast's, so it is checked by
                 properties: the sequences reconstruct the input exactly, frames decode bit-exactly under the oracle
                 decoder (= the reference's Decompressor), output is deterministic, ratio stays near the oracle's.
"""
import ctypes
import hashlib

import numpy as np
import pytest

import datagen
import zstdsharp_amd as z
from zstdsharp_amd import _ffi
from zstdsharp_amd.errors import ZSTD_ErrorCode, ZstdException

pytestmark = pytest.mark.gpu

SIZES = [0, 1, 2, 6, 7, 8, 9, 63, 64, 255, 256, 257, 1000, 4096, 16384, 65535, 65536, 65537, 131072, 200001]


@pytest.fixture(scope="module")
def ctxs(gpu_lib):
    c, d = z.Compressor(1), z.Decompressor()
    yield c, d
    c.Dispose(); d.Dispose()


@pytest.fixture(scope="module", params=[1, 2, 3], ids=["serial-literals", "selfsync-literals", "compact-literals"])
def forced_decoder(gpu_lib, request):
    """Every literal decoder (the default picks one by frame count): 1 = 4 lanes per frame with 4 KiB tables, 2 = 256 lanes
    per frame, 3 = 4 lanes per frame with compact (2 KiB + pairs) tables."""
    d = z.Decompressor()
    assert gpu_lib.ZSTDMI_DCtx_setLiteralDecoder(d.dctx, request.param) == 0
    yield d
    d.Dispose()


@pytest.mark.parametrize("kind", datagen.KINDS)
def test_round_trip_matrix(ctxs, oracle, kind):
    c, d = ctxs
    for n in SIZES:
        data = datagen.gen(kind, n, n + 1)
        comp = c.Wrap(data)
        assert len(comp) <= c.GetCompressBound(n)
        assert oracle.decompress(comp, n) == data, (kind, n, "oracle decode of GPU frames")
        assert z.Decompressor.GetDecompressedSize(comp) == n
        assert d.Unwrap(comp) == data, (kind, n, "GPU decode of GPU frames")
        ref = oracle.compress(data, 1, 0, 65536)
        assert d.Unwrap(ref) == data, (kind, n, "GPU decode of oracle frames")


def test_gpu_decoder_on_golden_frames(ctxs, golden):
    """Frames made by libzstd at levels 1..19 (multi-block, repeat tables, treeless literals, RLE/raw blocks, checksum,
    multi-frame + skippable frame): GPU output must be bit-exact."""
    _, d = ctxs
    for c in golden:
        blob = open(c["path"], "rb").read()
        out = d.Unwrap(blob)
        assert hashlib.sha256(out).hexdigest() == c["sha256"], c["file"]


def test_both_literal_decoders_on_golden_and_oracle_frames(forced_decoder, ctxs, oracle, golden):
    d = forced_decoder
    for c in golden:
        out = d.Unwrap(open(c["path"], "rb").read())
        assert hashlib.sha256(out).hexdigest() == c["sha256"], c["file"]
    comp = ctxs[0]
    for kind in datagen.KINDS:
        for n in (1, 255, 256, 1000, 65536, 65537, 200001):
            data = datagen.gen(kind, n, n + 3)
            assert d.Unwrap(comp.Wrap(data)) == data, (kind, n)
            for level in (1, 3, 5):
                ref = oracle.compress(data, level, 1, 65536)
                if isinstance(ref, int):
                    assert ref == -40 and level == 5          # the oracle refuses greedy/lazy without the row hash (windowLog <= 14)
                    continue
                assert d.Unwrap(ref) == data, (kind, n, level)
            assert d.Unwrap(oracle.compress(data, 1, 0, 0)) == data, (kind, n, "one multi-block frame")


def test_generate_buffer_sizes_with_reused_contexts(ctxs):
    """T/ZstdNetTests.cs:478-496: one Compressor/Decompressor pair reused over sizes 2, 3002, ... 99002."""
    c, d = ctxs
    for n in range(2, 100000, 3000):
        data = datagen.gen("bytei", n)
        assert d.Unwrap(c.Wrap(data)) == data


def test_header_known_answers(ctxs):
    """T/ZstdNetTests.cs:179-212 + SURVEY.md §8 a-2."""
    c, _ = ctxs
    comp = c.Wrap(datagen.gen("text", 1000, 1))
    assert comp[:4] == bytes([0x28, 0xB5, 0x2F, 0xFD]) and comp[4] == 0x60
    assert int.from_bytes(comp[5:7], "little") == 1000 - 256
    comp = c.Wrap(datagen.gen("zipf", 65536, 2))
    assert comp[:7] == bytes([0x28, 0xB5, 0x2F, 0xFD, 0x60, 0x00, 0xFF])
    assert c.Wrap(b"") == bytes([0x28, 0xB5, 0x2F, 0xFD, 0x20, 0x00, 0x01, 0x00, 0x00])


def test_checksum_flag(gpu_lib, oracle):
    """T/ZstdNetTests.cs:41-73: +4 bytes per frame, verified by the decoder; a flipped checksum is reported."""
    data = datagen.gen("text", 150000, 5)                       # 3 chunks -> 3 frames
    c0, c1, d = z.Compressor(1), z.Compressor(1), z.Decompressor()
    c1.SetParameter(201, 1)
    a, b = c0.Wrap(data), c1.Wrap(data)
    assert len(b) == len(a) + 4 * 3
    assert oracle.decompress(b, len(data)) == data
    assert d.Unwrap(b) == data
    bad = bytearray(b); bad[-1] ^= 0x40
    with pytest.raises(ZstdException) as e:
        d.Unwrap(bytes(bad))
    assert e.value.Code == ZSTD_ErrorCode.ZSTD_error_checksum_wrong
    small = datagen.gen("text", 5000, 6)
    assert len(c1.Wrap(small)) == len(c0.Wrap(small)) + 4      # the reference's own assertion (single frame)
    assert c1.Wrap(b"") == oracle.compress(b"", 1, 1)


def test_error_behaviour(ctxs, oracle):
    """T/ZstdNetTests.cs:166-258, 399-454."""
    c, d = ctxs
    data = datagen.gen("text", 5000, 7)
    comp = c.Wrap(data)
    with pytest.raises(ZstdException):
        d.Unwrap(bytes(range(1, 40)))                            # garbage
    ok, n = d.TryUnwrap(comp, bytearray(20))
    assert ok is False                                           # small destination -> soft failure
    with pytest.raises(ZstdException) as e:
        d.Unwrap(comp, bytearray(20))
    assert e.value.Code == ZSTD_ErrorCode.ZSTD_error_dstSize_tooSmall
    with pytest.raises(ZstdException) as e:
        d.Unwrap(comp, maxDecompressedSize=20)
    assert e.value.Code == ZSTD_ErrorCode.ZSTD_error_dstSize_tooSmall
    ok, n = c.TryWrap(datagen.gen("rand", 5000, 8), bytearray(100))
    assert ok is False
    with pytest.raises(ZstdException) as e:
        c.Wrap(datagen.gen("rand", 5000, 8), bytearray(100))
    assert e.value.Code == ZSTD_ErrorCode.ZSTD_error_dstSize_tooSmall
    with pytest.raises(ZstdException):
        d.Unwrap(comp[:-5])                                      # truncated
    with pytest.raises(ZstdException):
        d.Unwrap(comp + b"\x01\x02")                             # trailing garbage
    flipped = bytearray(comp); flipped[len(comp) // 2] ^= 0xFF
    want = oracle.decompress(bytes(flipped), len(data) + 64)     # corrupt payload: whatever the reference's decoder makes of it
    try:
        out = d.Unwrap(bytes(flipped), bytearray(len(data) + 64))
        assert not isinstance(want, int) and out == len(want), "the GPU decoder accepted a frame the oracle rejects"
    except ZstdException:
        assert isinstance(want, int), "the GPU decoder rejected a frame the oracle accepts"
    dest = bytearray(len(data) + 10)
    assert d.Unwrap(comp, dest, 10) == len(data) and bytes(dest[10:]) == data     # offset overload (T/ZstdNetTests.cs:260-397)


def _seqs_to_list(arr, n):
    return [(arr[i].offBase, arr[i].litLength, arr[i].mlBase) for i in range(n)]


def _gpu_chunk(lib, cctx, idx):
    seqs = (_ffi.ZSTDMI_Seq * 16384)()
    lits = ctypes.create_string_buffer(65536)
    ns, ls = ctypes.c_size_t(0), ctypes.c_size_t(0)
    r = lib.ZSTDMI_debugGetChunk(cctx, idx, seqs, 16384, ctypes.byref(ns), lits, 65536, ctypes.byref(ls))
    assert r == 0
    return _seqs_to_list(seqs, ns.value), lits.raw[:ls.value]


def _replay(seqs, lits, n, history=b"", rep=(1, 4, 8)):
    """Execute sequences the way the decoder does (repcode history included) -> reconstructed bytes.  history: what the frame
    holds in front of this block (a later block of a multi-block frame); rep: the repcodes the block may rely on (a later block
    is encoded against unknown ones, so any value must do: (0, 0, 0) makes a reliance fail the offset check)."""
    out = bytearray(history); rep = list(rep); lp = 0; n += len(history)
    for off_base, ll, mlb in seqs:
        out += lits[lp:lp + ll]; lp += ll
        ll0 = 1 if ll == 0 else 0
        if off_base > 3:
            off = off_base - 3; rep = [off, rep[0], rep[1]]
        else:
            idx = off_base - 1 + ll0
            if idx == 0:
                off = rep[0]
            else:
                off = rep[0] - 1 if idx == 3 else rep[idx]
                assert off != 0
                rep = [off, rep[0], rep[1]] if idx != 1 else [off, rep[0], rep[2]]
        ml = mlb + 3
        assert 0 < off <= len(out)
        for _ in range(ml):
            out.append(out[-off])
    out += lits[lp:]
    assert len(out) == n
    return bytes(out[len(history):])


@pytest.mark.parametrize("kind", ["text", "zipf", "runs", "mixed", "period", "zeros", "bytei"])
def test_match_finder_sequences_reconstruct_input(gpu_lib, ctxs, kind):
    """Row a-4: the GPU parse is not ZSTD_fast's parse, so check what any valid parse must satisfy."""
    c, _ = ctxs
    data = datagen.gen(kind, 65536 + 30000, 3)
    assert gpu_lib.ZSTDMI_CCtx_setHistory(c.cctx, 0, 0) == 0       # independent 64 KiB chunks (a call this small would get 16 KiB blocks, tests/test_gpu_history.py)
    try:
        c.Wrap(data)
        for idx, (lo, hi) in enumerate([(0, 65536), (65536, len(data))]):
            seqs, lits = _gpu_chunk(gpu_lib, c.cctx, idx)
            assert _replay(seqs, lits, hi - lo) == data[lo:hi]
            assert all(mlb + 3 >= 4 for _, _, mlb in seqs)
    finally:
        assert gpu_lib.ZSTDMI_CCtx_setHistory(c.cctx, -1, 0) == 0


@pytest.mark.parametrize("kind", ["text", "zipf", "runs", "mixed", "period", "zeros", "rand", "bytei"])
def test_entropy_stage_is_byte_identical_to_oracle(gpu_lib, ctxs, oracle, kind):
    """Rows a-7..a-11: same seqStore in -> same literals + sequences sections out, byte for byte."""
    c, _ = ctxs
    for n in (300, 5000, 20000, 65536):
        data = datagen.gen(kind, n, 9)
        # (i) the oracle's own ZSTD_fast parse, (ii) the GPU match finder's parse
        stores = [oracle.block_sequences(data, 1)]
        c.Wrap(data)
        stores.append(_gpu_chunk(gpu_lib, c.cctx, 0))
        for seqs, lits in stores:
            want = oracle.entropy_block(seqs, lits, n, 1)
            arr = (_ffi.ZSTDMI_Seq * max(len(seqs), 1))()
            for i, (o_, l_, m_) in enumerate(seqs):
                arr[i].offBase, arr[i].litLength, arr[i].mlBase = o_, l_, m_
            out = ctypes.create_string_buffer(n + 1024)
            r = gpu_lib.ZSTDMI_debugEntropyBlock(c.cctx, out, n + 1024, arr, len(seqs), lits, len(lits), n)
            assert r < (1 << 63), r
            assert out.raw[:r] == want, (kind, n, len(seqs), len(lits))


def test_output_is_deterministic(ctxs):
    c, _ = ctxs
    data = datagen.gen("mixed", 1 << 20, 4)
    first = c.Wrap(data)
    for _ in range(3):
        assert c.Wrap(data) == first


def test_ratio_stays_near_the_reference_parse(ctxs, oracle):
    """The wavefront-parallel parse may lose a little against ZSTD_fast's serial parse, not a lot."""
    c, _ = ctxs
    # slack = measured on MI355X + 2 % (round 2, with the region parse on dense chunks: zipf 1.0000, text 1.0154, runs 1.6608 —
    # runs of 50-400 equal bytes come out of the 64-byte regions in pieces where the lanes' stretches do not line up; the frames
    # are 2.5 % of the input either way —, mixed, bytei and period as printed)
    # Both at the same framing: independent single-block 64 KiB frames (ZSTDMI_CCtx_setHistory(0): a call of 1 MiB left to itself
    # gets four 16 KiB blocks per frame, whose price is pinned in tests/test_gpu_history.py)
    import zstdsharp_amd._ffi as _f
    lib = _f.load()
    assert lib.ZSTDMI_CCtx_setHistory(c.cctx, 0, 0) == 0
    try:
        for kind, slack in (("zipf", 1.02), ("text", 1.036), ("runs", 1.70), ("mixed", 1.03), ("bytei", 1.02), ("period", 1.02)):
            data = datagen.gen(kind, 1 << 20, 6)
            gpu, ref = len(c.Wrap(data)), len(oracle.compress(data, 1, 0, 65536))
            print(f"ratio-vs-oracle L1 {kind}: gpu {gpu} ref {ref} = {gpu / ref:.4f}")
            assert gpu <= ref * slack + 64, (kind, gpu, ref)
    finally:
        assert lib.ZSTDMI_CCtx_setHistory(c.cctx, -1, 0) == 0


def test_device_resident_api_and_large_input(gpu_lib, oracle):
    """BASELINE config 2 at reduced size (256 MiB of Zipf bytes, HBM resident) + size-independent properties:
    decompressBound == n, GPU decode == input, sha256 of oracle-decoded sample chunks."""
    import torch
    n = 256 << 20
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
    p = torch.arange(1, 257, dtype=torch.float64, device="cuda") ** -1.1
    cdf = torch.cumsum(p / p.sum(), 0).float()
    src = torch.searchsorted(cdf, torch.rand(n, device="cuda", generator=gen)).clamp_(max=255).to(torch.uint8)
    cap = gpu_lib.ZSTD_compressBound(n)
    dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    c, d = z.Compressor(1), z.Decompressor()
    torch.cuda.synchronize()
    csize = gpu_lib.ZSTDMI_compressDevice(c.cctx, dst.data_ptr(), cap, src.data_ptr(), n)
    assert csize < (1 << 63)
    assert 0.70 < csize / n < 0.74                                # order-0 entropy bound 0.7208
    r = gpu_lib.ZSTDMI_decompressDevice(d.dctx, back.data_ptr(), n, dst.data_ptr(), csize)
    assert r == n
    assert torch.equal(src, back)
    head = dst[:4 << 20].cpu().numpy().tobytes()                  # first frames, checked by the oracle decoder
    used, out = 0, bytearray()
    while used + 70000 < len(head):
        fs = oracle.lib().zso_findFrameCompressedSize(head[used:], len(head) - used)
        out += oracle.decompress(head[used:used + fs], 65536); used += fs
    assert bytes(out) == src[:len(out)].cpu().numpy().tobytes()
    c.Dispose(); d.Dispose()


def test_multi_pass_inputs(gpu_lib, oracle):
    """Inputs beyond one pass of the HBM workspace are compressed pass by pass; force 3-chunk passes on a 10-chunk input."""
    data = datagen.gen("mixed", 10 * 65536 + 777, 12)
    with z.Compressor(1) as c, z.Decompressor() as d:
        whole = c.Wrap(data)
        assert gpu_lib.ZSTDMI_CCtx_setPassChunks(c.cctx, 3) == 0
        comp = c.Wrap(data)
        assert comp == whole                                   # pass boundaries do not change the stream
        assert oracle.decompress(comp, len(data)) == data and d.Unwrap(comp) == data
        ok, _ = c.TryWrap(data, bytearray(len(comp) - 1))      # too small by one byte, detected in the last pass
        assert ok is False


def test_many_contexts_concurrently(gpu_lib):
    """T/ZstdNetTests.cs:498-522: many tasks, each with its own contexts."""
    import concurrent.futures as cf

    def work(i):
        data = datagen.gen("text", 20000 + 1000 * i, i)
        with z.Compressor(1) as c, z.Decompressor() as d:
            return d.Unwrap(c.Wrap(data)) == data
    with cf.ThreadPoolExecutor(8) as ex:
        assert all(ex.map(work, range(24)))


# ---------------------------------------------------------------------------------------------------------------
# rows a-5 / a-6: the higher-level match finders (dual-hash for levels 3-5, + lazy deferral for levels >= 6)
# ---------------------------------------------------------------------------------------------------------------
LEVEL_KINDS = ["text", "zipf", "runs", "mixed", "period", "zeros", "rand", "bytei"]


@pytest.mark.parametrize("level", [2, 3, 4, 5, 6, 9, 19])
def test_levels_round_trip(gpu_lib, oracle, level):
    """Every level's frames decode bit-exactly with the oracle (i.e. with the reference's decoder) and with the GPU decoder."""
    with z.Compressor(level) as c, z.Decompressor() as d:
        for kind in LEVEL_KINDS:
            for n in (1, 9, 300, 4097, 65536, 65537, 200001):
                data = datagen.gen(kind, n, n + level)
                comp = c.Wrap(data)
                assert len(comp) <= c.GetCompressBound(n)
                assert oracle.decompress(comp, n) == data, (level, kind, n)
                assert d.Unwrap(comp) == data, (level, kind, n)


@pytest.mark.parametrize("level", [3, 5, 7])
def test_level_finders_sequences_reconstruct_input(gpu_lib, level):
    with z.Compressor(level) as c:
        for kind in ("text", "mixed", "runs", "period"):
            data = datagen.gen(kind, 65536 + 30000, 3)
            assert gpu_lib.ZSTDMI_CCtx_setHistory(c.cctx, 0, 0) == 0          # independent 64 KiB chunks
            c.Wrap(data)
            for idx, (lo, hi) in enumerate([(0, 65536), (65536, len(data))]):
                seqs, lits = _gpu_chunk(gpu_lib, c.cctx, idx)
                assert _replay(seqs, lits, hi - lo) == data[lo:hi]
                assert all(mlb + 3 >= 4 for _, _, mlb in seqs)
            # the level's default: blocks of 48 / 32 KiB that match into the 16 / 32 KiB in front of them (row f-1); a later block's
            # sequences replay against the frame so far and never lean on repcodes they did not define themselves
            assert gpu_lib.ZSTDMI_CCtx_setHistory(c.cctx, -1, 0) == 0
            c.Wrap(data)
            reach = 0
            bs = 49152 if level < 5 else 32768            # doubleFast levels: 48 KiB blocks behind 16 KiB of history; above: 32 + 32
            for idx, lo in enumerate(range(0, len(data), bs)):
                hi = min(lo + bs, len(data))
                seqs, lits = _gpu_chunk(gpu_lib, c.cctx, idx)
                assert _replay(seqs, lits, hi - lo, history=data[:lo], rep=(1, 4, 8) if idx == 0 else (0, 0, 0)) == data[lo:hi]
                pos = 0
                for off_base, ll, mlb in seqs:
                    pos += ll
                    if off_base > 3: reach = max(reach, off_base - 3 - pos)
                    pos += mlb + 3
            assert reach <= 65536 - bs, "history is the 16 / 32 KiB in front of the block"
            if kind in ("text", "period"): assert reach > 0, "some match must reach in front of its block"


def test_levels_buy_ratio(gpu_lib, oracle):
    """More search effort must not lose ratio (small tolerance: the finders are heuristics), and each tier stays within a stated
    distance of the reference's own result for that level — at the same framing, and as the reference frames the input (one frame)."""
    sizes = {}
    for kind in ("text", "mixed", "bytei"):
        data = datagen.gen(kind, 1 << 20, 21)
        for level in (1, 3, 5, 7):
            with z.Compressor(level) as c:
                a = c.Wrap(data); b = c.Wrap(data)
                assert a == b                                   # deterministic at every level
                sizes[(kind, level)] = len(a)
        assert sizes[(kind, 3)] <= sizes[(kind, 1)] * 1.005, (kind, sizes)
        assert sizes[(kind, 5)] <= sizes[(kind, 3)] * 1.005, (kind, sizes)
        assert sizes[(kind, 7)] <= sizes[(kind, 5)] * 1.005, (kind, sizes)
        if kind == "text":     # level 5 earns its level: the hash-chain search (8 attempts) against the dual-hash finder of level 3
            assert sizes[(kind, 5)] <= sizes[(kind, 3)] * 0.97, (kind, sizes)
        # Against the reference's own parse (the oracle), at EQUAL framing — the level's frame span as independent frames: 240 KiB
        # (five 48 KiB blocks) at level 3, 256 KiB at level 5; inside a frame the oracle matches across the whole frame, the GPU
        # finders across 16 / 32 KiB of history plus the block (slack = measured + 2 %) —
        # and against what the reference really emits for this input, ONE frame with the level's full window (U/Clevels.cs:22, 62:
        # 512 KiB at level 1, 2 MiB at level 5): that gap is the window the LDS-resident finders do not have
        for level, span, slack, slack1 in ((3, 5 * (48 << 10), 1.10, 1.20), (5, 256 << 10, 1.085, 1.135)):   # round 3, 1 MiB of text: 1.081 / 1.180 and 1.064 / 1.112 (+ 2 %)
            ref = len(oracle.compress(data, level, 0, span))
            one = len(oracle.compress(data, level, 0, 0))
            print(f"ratio-vs-oracle L{level} {kind}: gpu {sizes[(kind, level)]} oracle at {span >> 10} KiB frames {ref} = {sizes[(kind, level)] / ref:.4f}, "
                  f"oracle one frame {one} = {sizes[(kind, level)] / one:.4f}")
            if kind != "bytei":      # (bytei: a 256-byte period — every finder collapses it to a few hundred bytes, ratios of tiny numbers)
                assert sizes[(kind, level)] <= ref * slack + 64, (kind, level, sizes[(kind, level)], ref)
                assert sizes[(kind, level)] <= one * slack1 + 64, (kind, level, sizes[(kind, level)], one)
        if kind != "bytei":
            one1 = len(oracle.compress(data, 1, 0, 0))
            print(f"ratio-vs-oracle L1 {kind}: gpu {sizes[(kind, 1)]} oracle one frame {one1} = {sizes[(kind, 1)] / one1:.4f}")
            assert sizes[(kind, 1)] <= one1 * 1.105 + 64, (kind, sizes[(kind, 1)], one1)
    print("level sizes", sizes)


# ---------------------------------------------------------------------------------------------------------------
# frames without a content size (what the reference's CompressionStream writes when no size is pledged,
# U/ZstdCompress.cs:4817-4929 with contentSizeFlag effectively off): decoded into bound-sized slots, then compacted
# ---------------------------------------------------------------------------------------------------------------
def _strip_content_size(frame: bytes, content_len: int) -> bytes:
    """Rewrite ONE frame's header so that it carries a window descriptor instead of a content size."""
    assert frame[:4] == b"\x28\xb5\x2f\xfd"
    fhd = frame[4]
    single, fcs_id, chk, did = (fhd >> 5) & 1, fhd >> 6, fhd & 4, fhd & 3
    assert did == 0
    old = 5 + (0 if single else 1) + [1 if single else 0, 2, 4, 8][fcs_id]
    wlog = max(10, (max(content_len, 1) - 1).bit_length())
    return frame[:4] + bytes([chk, (wlog - 10) << 3]) + frame[old:]


def test_frames_without_content_size(gpu_lib, oracle, forced_decoder):
    d = forced_decoder
    blobs, want = [], b""
    for kind, n in (("text", 200001), ("zipf", 65536), ("runs", 300), ("mixed", 400000), ("rand", 1000)):
        data = datagen.gen(kind, n, n)
        sized = oracle.compress(data, 1, 1 if kind == "mixed" else 0, 0)          # one (multi-block) frame, with checksum once
        unsized = _strip_content_size(sized, n)
        assert oracle.decompress(unsized, n) == data                               # the rewritten frame is valid zstd
        assert gpu_lib.ZSTD_getFrameContentSize(unsized, len(unsized)) == (1 << 64) - 1      # ZSTD_CONTENTSIZE_UNKNOWN
        bound = gpu_lib.ZSTD_decompressBound(unsized, len(unsized))
        assert n <= bound < (1 << 62)
        dest = bytearray(bound)
        assert d.Unwrap(unsized, dest) == n and bytes(dest[:n]) == data
        # Unwrap(src) sizes its buffer with the bound and returns the regenerated bytes (S/Decompressor.cs:63-75)
        assert d.Unwrap(unsized) == data
        # a destination of exactly the regenerated size is enough (ZSTD_decompressDCtx fails only if the bytes really overflow);
        # one byte less is dstSize_tooSmall, softly through TryUnwrap
        exact = bytearray(n)
        assert d.Unwrap(unsized, exact) == n and bytes(exact) == data
        with pytest.raises(ZstdException) as e:
            d.Unwrap(unsized, bytearray(n - 1))
        assert e.value.Code == ZSTD_ErrorCode.ZSTD_error_dstSize_tooSmall
        assert d.TryUnwrap(unsized, bytearray(n - 1)) == (False, 0)
        blobs.append(unsized); want += data
    # several unsized frames and a sized one in one buffer: every frame lands right behind the regenerated bytes of the one before
    sized_tail = oracle.compress(b"tail" * 1000, 1, 0, 0)
    both = b"".join(blobs) + sized_tail
    dest = bytearray(gpu_lib.ZSTD_decompressBound(both, len(both)))
    got = d.Unwrap(both, dest)
    assert bytes(dest[:got]) == want + b"tail" * 1000
    tight = bytearray(len(want) + 4000)
    assert d.Unwrap(both, tight) == len(tight) and bytes(tight) == want + b"tail" * 1000
    # and through the streaming adapter
    import io
    from zstdsharp_amd.streams import DecompressionStream
    with DecompressionStream(io.BytesIO(b"".join(blobs)), 1000) as ds:
        assert ds.ReadToEnd(7777) == want


def test_corrupted_frames_fail_cleanly(gpu_lib, oracle, forced_decoder):
    """Bit flips, byte overwrites and truncations of valid frames: the decoder must return an error or some bytes, never
    fault or hang, and must agree with the oracle whenever the oracle accepts the damaged frame without a checksum error
    (the reference's decoder behaviour is the oracle's, U/ZstdDecompressBlock.cs validations)."""
    import random
    d = forced_decoder
    rng = random.Random(20240611)
    seeds = []
    for kind, n, level in (("text", 3000, 1), ("zipf", 9000, 1), ("mixed", 70000, 1), ("runs", 2000, 1), ("text", 20000, 3)):
        data = datagen.gen(kind, n, n)
        seeds.append((oracle.compress(data, level, 1, 65536), len(data)))      # with checksum: silent corruption is detectable
        seeds.append((oracle.compress(data, level, 0, 0), len(data)))
    agree = errors = lenient = 0
    for blob, n in seeds:
        for _ in range(40):
            b = bytearray(blob)
            mode = rng.randrange(3)
            if mode == 0:
                i = rng.randrange(len(b)); b[i] ^= 1 << rng.randrange(8)
            elif mode == 1:
                i = rng.randrange(len(b)); b[i] = rng.randrange(256)
            else:
                del b[rng.randrange(4, len(b)):]
            b = bytes(b)
            want = oracle.decompress(b, n + 4096)
            dest = bytearray(n + 4096)
            try:
                got = d.Unwrap(b, dest)
                out = bytes(dest[:got])
            except ZstdException:
                out = None
            if isinstance(want, int):
                if out is None:
                    errors += 1
                else:
                    lenient += 1        # accepted a frame the oracle rejects
            else:
                assert out == want, "the decoders disagree on a frame the oracle accepts"
                agree += 1
    # every frame the oracle accepts decoded to the same bytes (asserted above, `agree` of them); frames it rejects must be
    # rejected here too — the validations are the reference's (U/ZstdDecompressBlock.cs), so none may slip through
    assert errors > 100 and agree > 0 and lenient == 0, (errors, agree, lenient)


def test_reference_shaped_frames_decode_block_parallel_at_size(gpu_lib, oracle):
    """BASELINE configs[4] at full per-launch shape: 1 GiB of level-5 frames as the reference's own Compressor emits them (1 MiB
    per frame = 8 chained 128 KiB blocks: history, repcodes, repeat-mode tables and treeless literals cross the blocks), decoded
    with compressed input and output resident in HBM.  64 MiB distinct (oracle-built here), repeated 16 times; bit-exact against
    the input, plus a checksum-of-checksums property: every frame carries an XXH64 the decoder verifies on the device."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    unique, reps, fb = 64 << 20, 16, 1 << 20
    data = datagen.gen("mixed", unique, 77)
    with ThreadPoolExecutor(max_workers=16) as ex:
        frames = list(ex.map(lambda i: oracle.compress(data[i:i + fb], 5, 1, 0), range(0, unique, fb)))
    blob = b"".join(frames)
    multi = sum(1 for f in frames if f[4] >> 6 == 2)            # 4-byte content size = a frame above 64 KiB: several blocks
    assert multi == len(frames)
    comp = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).cuda().repeat(reps)
    want = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda().repeat(reps)
    out = torch.empty(unique * reps, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with z.Decompressor() as d:
        r = gpu_lib.ZSTDMI_decompressDevice(d.dctx, out.data_ptr(), out.numel(), comp.data_ptr(), comp.numel())
        assert r == unique * reps, gpu_lib.ZSTD_getErrorName(r)
        assert torch.equal(out, want)
        # one flipped payload byte in the LAST frame: that frame's checksum (or a validation before it) must catch it
        bad = comp.clone(); bad[comp.numel() - 50] ^= 0x5A
        r = gpu_lib.ZSTDMI_decompressDevice(d.dctx, out.data_ptr(), out.numel(), bad.data_ptr(), bad.numel())
        assert r > (1 << 63)
        # a destination one byte short is dstSize_tooSmall before anything is written
        r = gpu_lib.ZSTDMI_decompressDevice(d.dctx, out.data_ptr(), out.numel() - 1, comp.data_ptr(), comp.numel())
        assert r == (1 << 64) - 70


def test_level5_compress_at_size(gpu_lib, oracle):
    """BASELINE configs[2] at full per-launch shape: 1 GiB (mixed corpus stand-in, 64 MiB distinct x 16) compressed at level 5 with
    input and output resident in HBM; GPU decode restores it, and a 4 MiB slice of the output decodes under the oracle."""
    import torch
    unique, reps = 64 << 20, 16
    n = unique * reps
    data = datagen.gen("mixed", unique, 78)
    src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda().repeat(reps)
    cap = gpu_lib.ZSTD_compressBound(n)
    dst = torch.empty(cap, dtype=torch.uint8, device="cuda"); out = torch.empty(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with z.Compressor(5) as c, z.Decompressor() as d:
        cs = gpu_lib.ZSTDMI_compressDevice(c.cctx, dst.data_ptr(), cap, src.data_ptr(), n)
        assert cs < (1 << 63), gpu_lib.ZSTD_getErrorName(cs)
        cs1 = gpu_lib.ZSTDMI_compressDevice(c.cctx, dst.data_ptr(), cap, src.data_ptr(), unique)      # the first 64 MiB alone: same frames
        head = dst[:cs1].cpu().numpy().tobytes()
        cs = gpu_lib.ZSTDMI_compressDevice(c.cctx, dst.data_ptr(), cap, src.data_ptr(), n)
        assert dst[:cs1].cpu().numpy().tobytes() == head, "frames are independent: a prefix of the input (whole frames) gives a prefix of the output"
        assert gpu_lib.ZSTDMI_decompressDevice(d.dctx, out.data_ptr(), n, dst.data_ptr(), cs) == n
        assert torch.equal(out, src)
        assert cs < 0.47 * n
    k = gpu_lib.ZSTD_findFrameCompressedSize(head, len(head))
    end = 0
    for _ in range(16):                                              # the first 16 frames = 4 MiB (level 5: 256 KiB frames of eight 32 KiB blocks)
        end += gpu_lib.ZSTD_findFrameCompressedSize(head[end:], len(head) - end)
    assert k > 0 and oracle.decompress(head[:end], 4 << 20) == data[:4 << 20]


def test_decode_prebuilt_level5_frames_at_size(gpu_lib, oracle):
    """BASELINE configs[4] at reduced size: decompress-only of pre-built level-5 frames (built by the oracle = the
    reference's level-5 algorithm), bit-exact check.  Both framings: independent 64 KiB frames, and one long multi-block
    frame with history across blocks (its blocks decode in parallel; only the match execution walks them in order)."""
    data = datagen.gen("mixed", 24 << 20, 55)
    with z.Decompressor() as d:
        chunked = oracle.compress(data, 5, 1, 65536)
        assert hashlib.sha256(d.Unwrap(chunked)).hexdigest() == hashlib.sha256(data).hexdigest()
        part = data[:6 << 20]
        whole = oracle.compress(part, 5, 1, 0)
        assert z.Decompressor.GetDecompressedSize(whole) == len(part)
        assert d.Unwrap(whole) == part


def test_boundary_sizes_differential(gpu_lib, oracle, forced_decoder):
    """Sizes around tile (4096), chunk (65536) and pass boundaries, every finder, every data kind: GPU frames decode under the
    oracle and under both GPU literal decoders; oracle frames decode on the GPU."""
    import random
    rng = random.Random(77)
    d = forced_decoder
    sizes = set()
    for base in (4096, 8192, 61440, 65536, 131072, 196608):
        for delta in (-9, -8, -7, -1, 0, 1, 7, 8, 9):
            sizes.add(base + delta)
    sizes |= {rng.randrange(1, 300000) for _ in range(25)}
    comps = {lvl: z.Compressor(lvl) for lvl in (1, 3, 6)}
    try:
        for n in sorted(sizes):
            kind = rng.choice(datagen.KINDS)
            data = datagen.gen(kind, n, n)
            for lvl, c in comps.items():
                comp = c.Wrap(data)
                assert d.Unwrap(comp) == data, (kind, n, lvl)
                if lvl == 1 or n % 7 == 0:
                    assert oracle.decompress(comp, n) == data, (kind, n, lvl)
            assert d.Unwrap(oracle.compress(data, 1, n & 1, 65536)) == data, (kind, n)
    finally:
        for c in comps.values():
            c.Dispose()


def test_repeat_after_incompressible_data_is_found(gpu_lib, oracle):
    """The match finder probes sparsely where the previous tile found nothing (as ZSTD_fast's growing step does); a repeat
    of incompressible data at an arbitrary distance must still be found (the probe lattice is jittered for that)."""
    import numpy as np
    rng = np.random.default_rng(5)
    for shift in (0, 1, 2, 3, 5, 13):
        block = rng.integers(0, 256, 30000, dtype=np.uint8).tobytes()
        chunk = block + bytes(rng.integers(0, 256, 100 + shift, dtype=np.uint8)) + block
        data = chunk * 3 + block[:5000]
        for level in (1, 5):
            with z.Compressor(level) as c, z.Decompressor() as d:
                comp = c.Wrap(data)
                assert d.Unwrap(comp) == data
                ref = len(oracle.compress(data, 1, 0, 65536))
                assert len(comp) <= ref * 1.10 + 512, (shift, level, len(comp), ref, len(data))


def test_output_is_deterministic_on_sparse_and_dense_data_at_size(gpu_lib):
    """Same input, same bytes out, run after run — also where the match finder probes sparsely (Zipf, random) and where long
    matches span whole tiles (zeros, runs): every cross-lane decision is order-independent by construction."""
    import torch
    for kind in ("zipf", "rand", "mixed", "zeros", "runs"):
        data = datagen.gen(kind, 48 << 20, 31)
        src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        cap = gpu_lib.ZSTD_compressBound(len(data))
        outs = []
        for level in (1, 5):
            with z.Compressor(level) as c:
                ref = None
                for _ in range(4):
                    dst = torch.zeros(cap, dtype=torch.uint8, device="cuda")
                    torch.cuda.synchronize()
                    cs = gpu_lib.ZSTDMI_compressDevice(c.cctx, dst.data_ptr(), cap, src.data_ptr(), len(data))
                    assert cs < (1 << 63)
                    if ref is None:
                        ref = (cs, dst[:cs].clone())
                    else:
                        assert cs == ref[0] and bool(torch.equal(dst[:cs], ref[1])), (kind, level)


def test_rle_and_raw_blocks_above_the_block_size_limit_follow_the_reference(gpu_lib, oracle):
    """A block header's size field is 21 bits wide; the reference's one-shot decoder checks a COMPRESSED block against the 128 KiB
    limit (U/ZstdDecompressBlock.cs:3095-3098) but takes an RLE or raw block's size as it stands (ZSTD_setRleBlock / ZSTD_copyRawBlock,
    U/ZstdDecompress.cs:1004-1052; only the streaming decoder refuses, :2965+).  Hand-made frames; whatever the oracle does, the GPU does."""
    def frame(blocks, n):
        out = bytes([0x28, 0xB5, 0x2F, 0xFD, 0xA0]) + n.to_bytes(4, "little")          # single segment, 4-byte content size
        for i, (btype, size, payload) in enumerate(blocks):
            hdr = (1 if i == len(blocks) - 1 else 0) | (btype << 1) | (size << 3)
            out += hdr.to_bytes(3, "little") + payload
        return out
    raw = datagen.gen("text", 150000, 5)
    cases = [frame([(1, 200000, b"\x5a")], 200000),                                        # one RLE block of 200 000 bytes
             frame([(0, 150000, raw)], 150000),                                              # one raw block of 150 000 bytes
             frame([(1, 131072, b"a"), (1, 300000, b"b"), (0, 10, b"0123456789")], 131072 + 300000 + 10),
             frame([(1, 200000, b"\x5a")], 199999)]                                          # content size and blocks disagree
    with z.Decompressor() as d:
        for blob in cases:
            want = oracle.decompress(blob, 1 << 20)
            try:
                got = d.Unwrap(blob)
            except z.ZstdException as e:
                got = -int(e.Code)
            assert got == want or (isinstance(want, int) and isinstance(got, int)), (len(blob), want if isinstance(want, int) else len(want), got if isinstance(got, int) else len(got))
