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
    assert lib.picles_abi_version() == 3


def test_struct_layouts_match_header_sizes():
    # field-by-field mirrors of the C structs (x86-64 SysV): sizes as the compiler lays them out
    assert C.sizeof(K.PiclesGrid) == 48
    assert C.sizeof(K.PiclesPhys) == 11 * 8 + 6 * 4 + 8
    assert C.sizeof(K.PiclesOde) == 4 * 8 + 2 * 4 + 8 + 4 * 8
    assert C.sizeof(K.PiclesModel) == 8 + 5 * 8
    assert C.sizeof(K.PiclesCounters) == 8 * 8 + 8 + 8 + 8


def _header_structs():
    """{name: [(c_type, field, array_len)]} of the `typedef struct picles_* { ... }` blocks of include/picles_hip.h"""
    h = (ROOT / "include" / "picles_hip.h").read_text()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    out = {}
    for name, body in re.findall(r"typedef\s+struct\s+(picles_[a-z_]+)\s*\{(.*?)\}\s*\1\s*;", h, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"((?:const\s+)?[a-z0-9_]+(?:\s*\*)?)\s*(.*)", decl)
            ctype = m.group(1).replace(" *", "*").replace("const ", "")
            for item in m.group(2).split(","):
                item = item.strip()
                ptr = item.startswith("*")
                item = item.lstrip("* ")
                a = re.match(r"([A-Za-z0-9_]+)(?:\[(\d+)\])?$", item)
                fields.append((ctype + ("*" if ptr else ""), a.group(1), int(a.group(2)) if a.group(2) else 0))
        out[name] = fields
    return out


def test_julia_binding_structs_mirror_the_header():
    """PiCLESHip.jl cannot be executed here (no Julia): its `struct picles_*` blocks are held against the header field by
    field — order, names and types — as the ctypes structures are by construction of the tests above"""
    jl = (ROOT / "picles_amd" / "julia" / "PiCLESHip.jl").read_text()
    C2J = {"int32_t": "Int32", "int64_t": "Int64", "uint64_t": "UInt64", "double": "Float64", "int8_t*": "Ptr{Int8}"}
    H = _header_structs()
    blocks = re.findall(r"^struct\s+(picles_[a-z_]+)\n(.*?)^end", jl, flags=re.S | re.M)
    assert {n for n, _ in blocks} == {"picles_grid", "picles_phys", "picles_ode", "picles_model", "picles_counters"}
    for name, body in blocks:
        jf = [tuple(x.strip().split("::")) for line in body.splitlines() for x in line.split("#")[0].split(";") if x.strip()]
        want = [(f, (C2J[t] if not n else f"NTuple{{{n},{C2J[t]}}}")) for t, f, n in H[name]]
        assert jf == want, (name, jf, want)
    # and the ctypes mirrors: same field names in the same order
    for cls, name in ((K.PiclesGrid, "picles_grid"), (K.PiclesPhys, "picles_phys"), (K.PiclesOde, "picles_ode"),
                      (K.PiclesModel, "picles_model"), (K.PiclesCounters, "picles_counters"), (K.PiclesTiming, "picles_timing")):
        assert [f for f, _ in cls._fields_] == [f for _, f, _ in H[name]], name
    m = re.search(r"const PICLES_ABI_VERSION = Int32\((\d+)\)", jl)
    assert int(m.group(1)) == K.ABI_VERSION


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
