import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def golden():
    import json
    d = os.path.join(ROOT, "tests", "golden")
    man = json.load(open(os.path.join(d, "manifest.json")))
    for c in man["cases"]:
        c["path"] = os.path.join(d, c["file"])
    return man["cases"]


@pytest.fixture(scope="session")
def golden_dict():
    """libzstd-made dictionary / unsized / small-window fixtures (tests/golden/make_golden_dict.py); dictionaries as bytes"""
    import json
    d = os.path.join(ROOT, "tests", "golden")
    man = json.load(open(os.path.join(d, "manifest_dict.json")))
    for c in man["cases"]:
        c["blob"] = open(os.path.join(d, c["file"]), "rb").read()
        c["dict_bytes"] = open(os.path.join(d, c["dict"]), "rb").read() if c.get("dict") else None
    return man["cases"]


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library, bound through its C ABI.  Fails (not skips) if it cannot drive a GPU."""
    # torch first: the wheel bundles its own copy of the HIP runtime, and it cannot enumerate the GPU once the system
    # runtime behind libzstd_mi355x.so holds it; the other order works (it is also bench.py's order)
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    torch.zeros(1, device="cuda")
    import zstdsharp_amd
    lib = zstdsharp_amd._ffi.load()
    assert lib.ZSTDMI_deviceCount() > 0, "no gfx950 device visible: GPU tests need the HIP path, there is no fallback"
    return lib
