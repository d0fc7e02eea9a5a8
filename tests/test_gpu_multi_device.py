"""SURVEY.md section 8 e inside the C ABI: ZSTDMI_CCtx_setDevices / ZSTDMI_DCtx_setDevices deal a call's frames to one worker per
listed device.  One GPU is enough to test it: the same ordinal listed several times gives several workers (own streams, own
workspaces, own host threads) on that device.  The contract: the bytes written do not depend on the number of workers."""
import ctypes

import numpy as np
import pytest

import datagen
import zstdsharp_amd as z
from zstdsharp_amd.errors import ZstdException, get_error_code, is_error

pytestmark = pytest.mark.gpu


def set_devices(lib, ctx_ptr, devs, compress=True):
    arr = (ctypes.c_int * len(devs))(*devs)
    r = (lib.ZSTDMI_CCtx_setDevices if compress else lib.ZSTDMI_DCtx_setDevices)(ctx_ptr, arr, len(devs))
    assert not is_error(r), get_error_code(r)


@pytest.mark.parametrize("level", [1, 3, 5])
@pytest.mark.parametrize("kind,n", [("text", 5_000_017), ("mixed", 9 << 20), ("zipf", 6 << 20), ("text", 100_000), ("bytei", 70_000)])
def test_workers_write_the_bytes_one_device_writes(gpu_lib, oracle, level, kind, n):
    """levels 1 / 3 / 5 (independent chunks, 240 KiB and 256 KiB history frames), inputs above and below the sparse-input probe's
    4 MiB, mixed input whose plan has several ranges, inputs smaller than a share: 2, 3 and 5 workers against one."""
    data = datagen.gen(kind, n, 11)
    with z.Compressor(level) as c:
        c.SetParameter(201, 1)
        one = c.Wrap(data)
        for devs in ([0, 0], [0, 0, 0], [0] * 5):
            set_devices(gpu_lib, c.cctx, devs)
            assert c.Wrap(data) == one, (level, kind, len(devs))
        set_devices(gpu_lib, c.cctx, [0])
        assert c.Wrap(data) == one
    assert oracle.decompress(one, n) == data
    with z.Decompressor() as d:
        ref = d.Unwrap(one)
        assert ref == data
        for devs in ([0, 0], [0, 0, 0, 0]):
            set_devices(gpu_lib, d.dctx, devs, compress=False)
            assert d.Unwrap(one) == data


def test_workers_with_a_dictionary_device_buffers_and_unsized_frames(gpu_lib, oracle):
    import torch
    dic = oracle.make_dictionary(datagen.gen("text", 30000, 5), datagen.gen("text", 60000, 6), 777)
    data = datagen.gen("text", 3 << 20, 5)
    with z.Compressor(1) as c, z.Decompressor() as d:
        c.LoadDictionary(dic); d.LoadDictionary(dic)
        one = c.Wrap(data)
        set_devices(gpu_lib, c.cctx, [0, 0, 0])
        assert c.Wrap(data) == one
        set_devices(gpu_lib, d.dctx, [0, 0, 0], compress=False)
        assert d.Unwrap(one) == data
    # device pointers in and out; frames without a content size (their places follow from what the shares regenerate)
    with z.Compressor(5) as c, z.Decompressor() as d:
        c.SetParameter(200, 0)
        ref = c.Wrap(data)
        set_devices(gpu_lib, c.cctx, [0, 0])
        src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
        cap = gpu_lib.ZSTD_compressBound(len(data))
        dst = torch.empty(cap, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
        r = gpu_lib.ZSTDMI_compressDevice(c.cctx, dst.data_ptr(), cap, src.data_ptr(), len(data))
        assert not is_error(r), get_error_code(r)
        assert dst[:r].cpu().numpy().tobytes() == ref
        set_devices(gpu_lib, d.dctx, [0, 0, 0], compress=False)
        back = torch.empty(len(data), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
        r2 = gpu_lib.ZSTDMI_decompressDevice(d.dctx, back.data_ptr(), len(data), dst.data_ptr(), r)
        assert r2 == len(data), get_error_code(r2)
        assert back.cpu().numpy().tobytes() == data


def test_worker_errors_and_bounds(gpu_lib, oracle):
    data = datagen.gen("text", 2 << 20, 3)
    with z.Compressor(1) as c:
        arr = (ctypes.c_int * 2)(0, 99)
        assert is_error(gpu_lib.ZSTDMI_CCtx_setDevices(c.cctx, arr, 2)), "a device that is not there"
        set_devices(gpu_lib, c.cctx, [0, 0, 0])
        c.SetParameter(201, 1)                                    # checksums: damage cannot go unnoticed
        comp = c.Wrap(data)
        small = ctypes.create_string_buffer(1000)
        r = gpu_lib.ZSTD_compress2(c.cctx, small, 1000, data, len(data))
        assert is_error(r) and get_error_code(r) == 70            # dstSize_tooSmall, as from one device
    with z.Decompressor() as d:
        set_devices(gpu_lib, d.dctx, [0, 0, 0], compress=False)
        bad = bytearray(comp); bad[len(bad) // 2] ^= 0x55; bad[len(bad) // 2 + 1] ^= 0x33
        with pytest.raises(ZstdException):
            d.Unwrap(bytes(bad))
        ok, n = d.TryUnwrap(comp, bytearray(1000))
        assert not ok
        assert d.Unwrap(comp) == data


@pytest.mark.parametrize("level", [1, 5])
def test_rank_shards_concatenate_to_the_one_gpu_stream(gpu_lib, oracle, level):
    """The torch.distributed path (bench.py --gpus N, zstdsharp_amd/dist.py): rank r compresses shard_range(total, r, N, unit) by
    itself and the all-gather-v puts the outputs one behind the other.  With the unit a multiple of the level's frame span (16 spans
    where the sparse-input probe groups frames) that concatenation IS what one GPU writes for the whole input — same plan, same frames."""
    from zstdsharp_amd.dist import frame_span, shard_range
    data = datagen.gen("text", 21_000_001, 4)
    with z.Compressor(level) as c:
        whole = c.Wrap(data)
        for world in (2, 4):
            parts = []
            for r in range(world):
                lo, hi = shard_range(len(data), r, world, 16 * frame_span(level))
                parts.append(c.Wrap(data[lo:hi]))
            assert b"".join(parts) == whole, (level, world)
