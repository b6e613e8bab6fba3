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


def test_integration_guide_names_every_entry_point():
    """INTEGRATION.md is the maintainer's map from the reference's methods to the C ABI: no exported entry point without a line in it"""
    doc = (ROOT / "INTEGRATION.md").read_text()
    missing = [n for n in _declared() if not re.search(r"\b" + n + r"\b", doc)]
    assert not missing, missing


def test_library_exports_every_declared_symbol():
    lib = K.load()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.picles_abi_version() == K.ABI_VERSION == int(re.search(r"#define PICLES_ABI_VERSION (\d+)", (ROOT / "include" / "picles_hip.h").read_text()).group(1))


def test_struct_layouts_match_header_sizes():
    # field-by-field mirrors of the C structs (x86-64 SysV): sizes as the compiler lays them out
    assert C.sizeof(K.PiclesGrid) == 48
    assert C.sizeof(K.PiclesPhys) == 11 * 8 + 6 * 4 + 8
    assert C.sizeof(K.PiclesOde) == 4 * 8 + 2 * 4 + 8 + 4 * 8
    assert C.sizeof(K.PiclesModel) == 8 + 5 * 8
    assert C.sizeof(K.PiclesCounters) == 8 * 8 + 8 + 8 + 8
    assert C.sizeof(K.PiclesSlabPhases) == 2 * 8 + 5 * 8


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


def _header_prototypes():
    """name -> (return type, [argument types]) of every function include/picles_hip.h declares, C spelling normalised"""
    h = (ROOT / "include" / "picles_hip.h").read_text()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    h = re.sub(r"typedef struct \w+ \{.*?\} \w+;", "", h, flags=re.S)
    h = re.sub(r"^\s*#.*$", "", h, flags=re.M)          # preprocessor lines
    h = re.sub(r"typedef [^;]*;", "", h)
    out = {}
    for ret, name, args in re.findall(r"([\w\s\*]+?)\b(picles_[a-z_0-9]+)\s*\(([^;{}]*?)\)\s*;", h):
        ret = " ".join(ret.replace("*", " * ").split())
        al = []
        for a in args.split(","):
            a = " ".join(a.replace("*", " * ").split())
            if a in ("void", ""):
                continue
            toks = a.split()
            if toks[-1] != "*" and len(toks) > 1:       # drop the parameter name
                toks = toks[:-1]
            al.append(" ".join(toks))
        out[name] = (ret, al)
    return out


def _norm_c(t):
    t = t.replace("const ", "").strip()
    stars = t.count("*")
    base = t.replace("*", "").strip()
    base = {"picles_ctx": "void", "char": "char"}.get(base, base)
    base = {"int32_t": "i32", "int64_t": "i64", "uint8_t": "u8", "int8_t": "i8", "double": "f64", "size_t": "u64", "void": "void",
            "char": "char"}.get(base, base)
    return base + "*" * stars


def _norm_jl(t):
    t = t.strip()
    m = re.fullmatch(r"(?:Ptr|Ref)\{(.*)\}", t)
    if m:
        return _norm_jl(m.group(1)) + "*"
    return {"Int32": "i32", "Cint": "i32", "Int64": "i64", "UInt8": "u8", "Int8": "i8", "Float64": "f64", "Cdouble": "f64", "Csize_t": "u64",
            "Cvoid": "void", "Cstring": "char*"}.get(t, t)


def test_every_julia_ccall_matches_its_header_prototype():
    """the Julia shim cannot be executed here (no Julia in the image), so its 29 ccall sites are held against the
    header statically: symbol declared, return type, arity and every argument type (Ptr / Ref of the same pointee count as equal;
    picles_ctx* is Ptr{Cvoid}; Cstring is const char*)"""
    jl = (ROOT / "picles_amd" / "julia" / "PiCLESHip.jl").read_text()
    protos = _header_prototypes()
    assert set(protos) == set(_declared())
    sites = re.findall(r"ccall\(\(:(picles_\w+), libpicles\),\s*([\w{}]+),\s*\((.*?)\)\s*(?:,|\))", jl, flags=re.S)
    assert len(sites) == len(re.findall(r"ccall\(\(:picles_", jl)) >= 29          # every call site was parsed
    for name, ret, args in sites:
        assert name in protos, f"PiCLESHip.jl calls {name}, which the header does not declare"
        cret, cargs = protos[name]
        jargs = [a for a in (x.strip() for x in re.sub(r"\s+", " ", args).split(",")) if a]
        assert _norm_jl(ret) == _norm_c(cret), (name, ret, cret)
        assert len(jargs) == len(cargs), (name, jargs, cargs)
        for k, (ja, ca) in enumerate(zip(jargs, cargs)):
            assert _norm_jl(ja) == _norm_c(ca), (name, k, ja, ca)


def test_ctypes_binding_matches_header_prototypes():
    """the same for the ctypes table of the Python host layer (picles_amd/_capi.py SYMBOLS)"""
    protos = _header_prototypes()
    def norm_ct(t):
        if t is None:
            return "void"
        if t is C.c_char_p:
            return "char*"
        if t is C.c_void_p:
            return "void*"
        if hasattr(t, "_type_") and not isinstance(t._type_, str):     # POINTER(T)
            return norm_ct(t._type_) + "*"
        return {C.c_int32: "i32", C.c_int64: "i64", C.c_double: "f64", C.c_uint8: "u8", C.c_int8: "i8", C.c_size_t: "u64",
                C.c_uint64: "u64"}.get(t, getattr(t, "__name__", str(t)))
    structs = {"PiclesGrid": "picles_grid", "PiclesPhys": "picles_phys", "PiclesOde": "picles_ode", "PiclesModel": "picles_model",
               "PiclesCounters": "picles_counters", "PiclesTiming": "picles_timing", "PiclesSlabPhases": "picles_slab_phases"}
    for name, (res, args) in K.SYMBOLS.items():
        cret, cargs = protos[name]
        assert norm_ct(res) == _norm_c(cret), (name, res, cret)
        assert len(args) == len(cargs), (name, len(args), cargs)
        for k, (pa, ca) in enumerate(zip(args, cargs)):
            got, want = norm_ct(pa), _norm_c(ca)
            for py, cn in structs.items():
                got = got.replace(py, cn)
            # an opaque void* stands for any object pointer (ctx handles, id blobs, hipStream_t)
            assert got == want or (got == "void*" and want.endswith("*")) or (got == "void**" and want.endswith("**")), (name, k, got, want)
