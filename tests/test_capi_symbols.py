"""The C-ABI library loads, exports every symbol include/picles_hip.h declares, and refuses to
run without a HIP device (no CPU fallback).  No compute calls: runs on the CPU-only box."""
import ctypes as C
import re
from pathlib import Path

import pytest

from picles_amd import _capi as K, configs, models

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    h = (ROOT / "include" / "picles_hip.h").read_text()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(picles_[a-z_0-9]+)\s*\(", h)))


def test_header_and_binding_agree():
    assert _declared() == sorted(K.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = K.load()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.picles_abi_version() == 2


def test_struct_layouts_match_header_sizes():
    # field-by-field mirrors of the C structs (x86-64 SysV): sizes as the compiler lays them out
    assert C.sizeof(K.PiclesGrid) == 48
    assert C.sizeof(K.PiclesPhys) == 11 * 8 + 6 * 4 + 8
    assert C.sizeof(K.PiclesOde) == 4 * 8 + 2 * 4 + 8 + 4 * 8
    assert C.sizeof(K.PiclesModel) == 8 + 5 * 8
    assert C.sizeof(K.PiclesCounters) == 8 * 8 + 8 + 8


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    cfg = configs.example_00_minimal(n=9, L=16e3)
    with pytest.raises(K.PiclesError, match="no HIP device"):
        models.WaveGrowth2D(**cfg.model)


def test_product_never_imports_oracle():
    for f in (ROOT / "picles_amd").rglob("*"):
        if f.suffix in (".py", ".h", ".hip", ".cpp") :
            txt = f.read_text(errors="ignore")
            assert "_oracle" not in txt and "liboracle" not in txt and "oracle/" not in txt.replace("CPU oracle", ""), f
