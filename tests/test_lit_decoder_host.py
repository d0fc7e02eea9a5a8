"""CPU test of the literal decoders' inner function: huf_decode_stream_fs (decode_lit.hip) is plain integer code apart from
v_alignbit_b32, so its text is lifted out of the .hip file, compiled for the host with a software alignbit (AddressSanitizer
on: the GPU pool has none) and run over 30 000 random Huffman codes / streams in both table forms (11-bit index; 10-bit index
with paired 11-bit codes), including the short streams whose last steps read below the stream's start."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_stream_decoder_on_the_host(tmp_path):
    src = open(os.path.join(ROOT, "zstdsharp_amd", "csrc", "decode_lit.hip")).read()
    a = src.index("template <u32 IDX, bool PAIRS>\n__device__ __forceinline__ bool huf_decode_stream_fs")
    b = src.index("// ------------------------------------------------------------------------------------------------\n// literals, fast path")
    (tmp_path / "fs.inc").write_text(src[a:b].replace("__device__ __forceinline__", "static inline").replace("__restrict__", ""))
    hdr = open(os.path.join(ROOT, "zstdsharp_amd", "csrc", "zmi_decode.h")).read()
    bb = hdr[hdr.index("struct BackBits {"):hdr.index("struct SeqSym")].replace("__device__ __forceinline__", "inline")
    (tmp_path / "bb.inc").write_text(bb)
    shutil.copy(os.path.join(ROOT, "tests", "host", "lit_stream_harness.cpp"), tmp_path / "t.cpp")
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fsanitize=address", "-o", str(exe), str(tmp_path / "t.cpp")], cwd=tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True, cwd=tmp_path)
    assert out.returncode == 0 and "done bad=0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
