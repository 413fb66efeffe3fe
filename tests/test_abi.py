"""CPU-side checks of the C-ABI boundary: the library builds, loads and exports every symbol include/snpgpu.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "snpgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(snpgpu_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from snp_pipeline_amd import _lib
    assert set(_declared_symbols()) == set(_lib.SIGNATURES)


def test_library_builds_loads_and_exports_everything():
    from snp_pipeline_amd import build, _lib
    path = build.build(verbose=False)
    assert os.path.exists(path)
    lib = _lib.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.snpgpu_abi_version() == 7
    assert lib.snpgpu_packed_row_bytes(33) == 64 and lib.snpgpu_packed_row_bytes(128) == 64 and lib.snpgpu_packed_row_bytes(129) == 128   # rows padded to 4 words
    assert ctypes.sizeof(_lib.SiteCounts) == 128


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from snp_pipeline_amd import device
    with pytest.raises(device.SnpGpuError):
        device.Device(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "snp_pipeline_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn
