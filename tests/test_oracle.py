"""CPU tests: the oracle against the committed golden vectors and the reference tests' known answers.

The oracle is a restatement of the reference's algorithm (oracle/*.c).  It is pinned here by
  * libzstd-produced frames (tests/golden, see make_golden.py for why libzstd stands in for the reference),
  * the known answers of the reference's own tests (T/ZstdNetTests.cs), cited per test.
"""
import hashlib

import pytest

import datagen


def _input(case):
    return datagen.gen(case["kind"], case["n"], case["seed"]) * case["copies"]


def test_decoder_matches_every_golden_frame(oracle, golden):
    for c in golden:
        blob = open(c["path"], "rb").read()
        data = _input(c)
        assert hashlib.sha256(data).hexdigest() == c["sha256"], c["file"]       # the generator still regenerates the input
        assert oracle.lib().zso_decompressBound(blob, len(blob)) == len(data), c["file"]
        assert oracle.decompress(blob, len(data)) == data, c["file"]


# fixtures whose libzstd-1.5.7 bytes a faithful restatement of 1.5.1's level-1 path must reproduce exactly
# (single-block frames: nothing 1.5.2+ changed is reachable; checked when the fixtures were generated)
BYTE_IDENTICAL = ["bytei_0_l1", "bytei_1_l1", "bytei_2_l1", "bytei_3002_l1", "bytei_12002_l1", "bytei_99002_l1",
                  "text_20000_l1", "zipf_65536_l1", "runs_50000_l1", "rand_5000_l1", "period_9000_l1", "zeros_200000_l1"]


def test_encoder_reproduces_golden_level1_bytes(oracle, golden):
    by_name = {c["file"][:-4]: c for c in golden}
    for name in BYTE_IDENTICAL:
        c = by_name[name]
        assert oracle.compress(_input(c), c["level"], c["checksum"]) == open(c["path"], "rb").read(), name


@pytest.mark.parametrize("kind", datagen.KINDS)
def test_encoder_round_trip(oracle, kind):
    for n in (0, 1, 6, 7, 8, 255, 256, 4096, 65535, 65536, 65537, 140000, 400000):
        data = datagen.gen(kind, n, n)
        for chunk in (0, 65536):
            comp = oracle.compress(data, 1, 0, chunk)
            assert isinstance(comp, bytes)
            assert oracle.decompress(comp, n) == data, (kind, n, chunk)


def test_doublefast_levels_round_trip(oracle):
    """Levels 2-4 (doubleFast where U/Clevels.cs says so).  Pinned by round trips only: libzstd 1.4.8 predates the 1.5.1
    rewrite of this match finder and 1.5.7 changed it again, so no byte-identical stand-in exists here (131 of 168 swept
    cases do match 1.5.7 byte for byte)."""
    for kind in ("text", "mixed", "runs", "zipf"):
        for n in (20000, 131073, 300000):
            data = datagen.gen(kind, n, n)
            for level in (2, 3, 4):
                comp = oracle.compress(data, level)
                if comp == -40:        # this (level, size tier) is greedy in U/Clevels.cs: refused, not substituted
                    continue
                assert isinstance(comp, bytes) and oracle.decompress(comp, n) == data, (kind, n, level)


def test_generate_buffer_sizes_round_trip(oracle):
    """T/ZstdNetTests.cs:478-496: (byte)i buffers of 2, 3002, ... 99002 bytes."""
    for n in range(2, 100000, 3000):
        data = datagen.gen("bytei", n)
        assert oracle.decompress(oracle.compress(data, 1), n) == data


def test_frame_header_known_answers(oracle):
    """T/ZstdNetTests.cs:179-212: for a 256..65791-byte input, byte 4 is 0x60 (single segment, 2-byte FCS, no checksum,
    no dictID) and the FCS field sits at byte 5 holding size-256."""
    data = datagen.gen("text", 1000, 1)
    comp = oracle.compress(data, 1)
    assert comp[:4] == bytes([0x28, 0xB5, 0x2F, 0xFD])
    assert comp[4] == 0x60
    assert int.from_bytes(comp[5:7], "little") == 1000 - 256
    # 64 KiB chunk of the GPU framing: 28 B5 2F FD 60 00 FF
    comp = oracle.compress(datagen.gen("zipf", 65536, 2), 1)
    assert comp[:7] == bytes([0x28, 0xB5, 0x2F, 0xFD, 0x60, 0x00, 0xFF])


def test_checksum_adds_exactly_four_bytes(oracle):
    """T/ZstdNetTests.cs:41-73."""
    data = datagen.gen("text", 5000, 2)
    a, b = oracle.compress(data, 1, 0), oracle.compress(data, 1, 1)
    assert len(b) == len(a) + 4
    assert oracle.decompress(b, len(data)) == data
    corrupted = b[:-1] + bytes([b[-1] ^ 1])
    assert oracle.decompress(corrupted, len(data)) == -22          # checksum_wrong


def test_empty_and_one_byte(oracle):
    """T/ZstdNetTests.cs:456-476."""
    assert len(oracle.compress(b"", 1)) == 9
    assert oracle.decompress(oracle.compress(b"", 1), 0) == b""
    assert oracle.decompress(oracle.compress(bytes([42]), 1), 1) == bytes([42])


def test_error_behaviour(oracle):
    """T/ZstdNetTests.cs:166-258: garbage -> error; small destination -> dstSize_tooSmall (-70)."""
    assert oracle.decompress(bytes(range(1, 20)), 100) == -10              # prefix_unknown
    data = datagen.gen("text", 5000, 3)
    comp = oracle.compress(data, 1)
    assert oracle.decompress(comp, 20) == -70
    assert isinstance(oracle.decompress(comp[:-3], len(data)), int)        # truncated
    assert oracle.decompress(comp + b"\x00\x01", len(data)) == -72         # trailing garbage -> srcSize_wrong


def test_xxh64_known_answers(oracle):
    """XXH64 test vectors (seed 0) published with the xxHash specification."""
    assert oracle.lib().zso_xxh64(b"", 0, 0) == 0xEF46DB3751D8E999
    assert oracle.lib().zso_xxh64(b"a", 1, 0) == 0xD24EC4F1A98C6E5B
    assert oracle.lib().zso_xxh64(b"abc", 3, 0) == 0x44BC2CF5AD770999


def test_unrestated_strategies_are_refused_not_substituted(oracle):
    data = datagen.gen("text", 10000, 1)
    assert oracle.compress(data, 5) == -40          # <= 16 KiB at level 5 is lazy over the hash-chain finder: not restated


def test_greedy_row_hash_levels_round_trip(oracle):
    """Levels 4-5 where U/Clevels.cs selects greedy with windowLog > 14 (row-hash match finder, U/ZstdLazy.cs:1101-1309).
    Pinned by round trips; 71 of 112 swept cases are byte-identical to libzstd 1.5.7, the rest differ by a handful of bytes
    (1.5.2+ changed the row finder: hash salt, tag layout), so byte-identity is not asserted."""
    for kind in ("text", "mixed", "zipf", "runs"):
        for n in (20000, 65536, 300000):
            data = datagen.gen(kind, n, n)
            for level in (4, 5):
                for chunk in (0, 65536):
                    comp = oracle.compress(data, level, 0, chunk)
                    if comp == -40:
                        continue
                    assert isinstance(comp, bytes) and oracle.decompress(comp, n) == data, (kind, n, level, chunk)


def test_raw_content_dictionary_frames(oracle):
    """Row f-4's checker: the oracle's dictionary decoder (ZSTD_refDictContent + ZSTD_execSequence's extDict branch,
    U/ZstdDecompress.cs:1758-1771, U/ZstdDecompressBlock.cs:2223-2250) against its dictionary-frame generator.
    Parity unpinned for the frame BYTES (the reference holds no dictionary fixtures); pinned behaviour: a dictionary
    frame restores the data with the dictionary, fails without it (T/ZstdNetTests.cs:95-113), dictionaries under 8 bytes
    are ignored (U/ZstdCompress.cs:5469-5477), dictionary-less frames decode with any dictionary loaded (:76-93)."""
    import numpy as np
    r = np.random.default_rng(3)
    vocab = [bytes(r.integers(97, 123, size=int(r.integers(3, 10))).astype(np.uint8)) for _ in range(200)]

    def text(n, seed):
        g = np.random.default_rng(seed); out = bytearray()
        while len(out) < n:
            out += vocab[int(g.integers(0, len(vocab)))] + b" "
        return bytes(out[:n])

    dic = text(20000, 1)
    for n in (0, 1, 8, 100, 3000, 70000, 300000):
        data = text(n, n + 2)
        for chk in (0, 1):
            frame = oracle.compress_dict(data, dic, 1, chk)
            assert not isinstance(frame, int)
            assert oracle.decompress(frame, n, dic) == data
            if n >= 100:
                assert len(frame) < len(oracle.compress(data, 1, chk))          # the dictionary helps (T/ZstdNetTests.cs:148-164)
                assert oracle.decompress(frame, n) in (-20, -22)                 # corruption_detected / checksum_wrong without it
        assert oracle.decompress(oracle.compress(data, 1, 0), n, dic) == data
        assert oracle.compress_dict(data, b"1234567", 1, 0) == oracle.compress(data, 1, 0)
    assert oracle.compress_dict(b"x" * 100, bytes([0x37, 0xA4, 0x30, 0xEC]) + bytes(60), 1, 0) == -30    # the magic, then garbage: dictionary_corrupted
    assert oracle.compress_dict(text(5000, 9), dic, 3, 0) == -40                                           # fast strategy only


def test_formatted_dictionary_frames(oracle):
    """Formatted dictionaries (magic, dictID, Huffman + FSE tables, repcodes, content — ZSTD_loadZstdDictionary,
    U/ZstdCompress.cs:5402-5463; ZSTD_loadDEntropy, U/ZstdDecompress.cs:1773-1875), written by the oracle's test writer.
    Pinned behaviour: frame header byte 4 carries the dictID size code (0x63-style with a 4-byte id, T/ZstdNetTests.cs:
    179-212), decoding needs that very dictionary (dictionary_wrong otherwise, :95-134), and the dictionary's tables make
    tiny inputs smaller than raw content alone does.  Frame bytes: parity unpinned (no dictionary fixtures in the reference)."""
    import numpy as np
    r = np.random.default_rng(3)
    vocab = [bytes(r.integers(97, 123, size=int(r.integers(3, 10))).astype(np.uint8)) for _ in range(200)]

    def text(n, seed):
        g = np.random.default_rng(seed); out = bytearray()
        while len(out) < n:
            out += vocab[int(g.integers(0, len(vocab)))] + b" "
        return bytes(out[:n])

    content, sample = text(20000, 1), text(60000, 2)
    dic = oracle.make_dictionary(content, sample, 0x12345678)
    assert dic[:8] == bytes([0x37, 0xA4, 0x30, 0xEC, 0x78, 0x56, 0x34, 0x12]) and dic.endswith(content)
    other = oracle.make_dictionary(content, sample, 77)
    for n in (0, 1, 8, 100, 3000, 70000, 300000):
        data = text(n, n + 2)
        frame = oracle.compress_dict(data, dic, 1, 0)
        assert frame[4] & 3 == 3 and frame[4] & 0x20 and frame[5:9] == dic[4:8]     # single segment, 4-byte dictID right after the FHD
        assert oracle.decompress(frame, n, dic) == data
        assert oracle.decompress(frame, n) == -32 and oracle.decompress(frame, n, content) == -32 and oracle.decompress(frame, n, other) == -32
        if 100 <= n <= 3000:
            assert len(frame) < len(oracle.compress_dict(data, content, 1, 0)) < len(oracle.compress(data, 1, 0))
        assert oracle.decompress(oracle.compress(data, 1, 0), n, dic) == data        # dictionary-less frames decode with it loaded
    assert oracle.decompress(oracle.compress_dict(text(500, 5), dic, 1, 0), 500, dic[:40]) == -30      # truncated header: dictionary_corrupted


def test_decoder_on_libzstd_dictionary_and_unsized_frames(oracle, golden_dict):
    """Frames this repo did not make (tests/golden/make_golden_dict.py, libzstd 1.5.7): compressed against a ZDICT-trained
    dictionary (formatted: its Huffman / FSE tables and repcodes are in use, the frames name its dictID) and against a raw-content
    one, streamed without a pledged size (no content size in the header), and with windowLog 11 + checksum
    (T/ZstdNetSteamingTests.cs:293).  The oracle decoder must regenerate every input bit-exactly."""
    import hashlib
    assert len(golden_dict) == 16
    for c in golden_dict:
        out = oracle.decompress(c["blob"], c["n"] + 64, c["dict_bytes"])
        assert not isinstance(out, int), (c["file"], out)
        assert len(out) == c["n"] and hashlib.sha256(out).hexdigest() == c["sha256"], c["file"]
        if c.get("dict") == "trained_16k.dict":                      # these frames name the dictionary's ID: without it -> dictionary_wrong
            assert c["blob"][4] & 3 != 0 and oracle.decompress(c["blob"], c["n"] + 64) == -32, c["file"]
        if c.get("unsized"):
            fhd = c["blob"][4]
            assert fhd >> 6 == 0 and not (fhd >> 5) & 1, "no content size field, not single-segment"
