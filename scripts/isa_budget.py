"""Instruction budget of a k_step / k_advance flavour from the compiler's assembly: the depth-1 loop with the most fp64 products is the
Runge-Kutta loop; its body is split into basic blocks and every VALU / SALU / memory instruction is filed under a class.  The numbers
are STATIC counts per loop body (one RK attempt when every block of the body executes once; blocks behind rare branches are listed
separately so that they can be left out).

    python scripts/isa_budget.py k_step_explicit.hip _Z6k_stepILb1ELb0ELb1ELb0ELb0E        # the BASELINE kernel
    python scripts/isa_budget.py k_step_auto.hip     _Z6k_stepILb1ELb1ELb1ELb0ELb1E        # the default solver, static winds
"""
import re
import subprocess
import sys
from collections import Counter, OrderedDict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "picles_amd" / "csrc"
import os
FLAGS = os.environ.get("ISA_EXTRA", "").split() + ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-munsafe-fp-atomics", "-fPIC", "--cuda-device-only", "-w"]

CLASSES = OrderedDict([
    ("fp64 fma", r"v_fma_f64|v_fmac_f64"),
    ("fp64 mul", r"v_mul_f64"),
    ("fp64 add", r"v_add_f64"),
    ("fp64 min/max", r"v_(min|max)_f64"),
    ("fp64 rcp/rsq/sqrt", r"v_(rcp|rsq|sqrt)_f64"),
    ("fp64 rndne/ldexp/frexp/fract/trunc/floor", r"v_(rndne|ldexp|frexp_\w+|fract|trunc|floor|ceil)_f64"),
    ("fp64 cmp / class", r"v_cmp\w*_f64|v_cmpx\w*_f64|v_cmp_class_f64|v_div_\w+_f64"),
    ("cvt", r"v_cvt_"),
    ("mov (v_mov / accvgpr)", r"v_mov_b32|v_mov_b64|v_accvgpr|v_pk_mov"),
    ("select (v_cndmask)", r"v_cndmask"),
    ("lane moves (readlane / writelane / readfirstlane / dpp / permute)", r"v_readlane|v_writelane|v_readfirstlane|_dpp|ds_bpermute|ds_permute|v_permlane"),
    ("int / logic VALU", r"v_(and|or|xor|not|lshl|lshr|ashr|add_u|add_co|addc|sub_u|sub_co|subb|add3|mad_u|mad_i|mul_lo|mul_hi|bfe|bfi|lshl_add|add_lshl|and_or|or3|min_[iu]|max_[iu]|cmp\w*_[iu]\d|cmpx\w*_[iu]\d|sub_nc|add_nc|mul_u|mul_i|alignbit|perm_b32|bcnt|mbcnt|ffb|med3)"),
    ("scratch (spill traffic)", r"scratch_"),
    ("LDS", r"ds_(read|write|load|store|add|min|max|swizzle)"),
    ("global / flat memory", r"global_|flat_|buffer_"),
    ("scalar memory (s_load)", r"s_load|s_buffer_load"),
    ("SALU", r"s_(?!waitcnt|nop|load|buffer_load|branch|cbranch|endpgm|barrier|sleep|setprio|sendmsg|delay|clause)"),
    ("branch", r"s_branch|s_cbranch"),
    ("waitcnt / nop", r"s_waitcnt|s_nop|s_delay|s_sleep"),
])


def classify(ins):
    for name, pat in CLASSES.items():
        if re.match(r"(" + pat + r")", ins):
            return name
    return "other:" + ins.split()[0]


_ASM = {}


def assembly(unit):
    """the compiler's assembly of a translation unit (compiled once per process)"""
    if unit not in _ASM:
        _ASM[unit] = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-S", str(SRC / unit), "-o", "-"], capture_output=True, text=True, cwd=SRC, timeout=1200).stdout
    return _ASM[unit]


def kernels(unit):
    return [m.group(1) for m in re.finditer(r"^(_Z\d+k_(?:step|advance)I\w*):\s*;", assembly(unit), re.M)]


def budget(unit, prefix, verbose=False):
    asm = assembly(unit)
    heads = list(re.finditer(r"^(_Z\d+k_(?:step|advance)I\w*):\s*;", asm, re.M))
    sel = [(m, nxt) for m, nxt in zip(heads, heads[1:] + [None]) if m.group(1).startswith(prefix)]
    assert len(sel) == 1, [m.group(1) for m, _ in sel]
    m, nxt = sel[0]
    body = asm[m.end():(nxt.start() if nxt else len(asm))].split(".Lfunc_end")[0].split("\n")
    # basic blocks with their loop header annotation
    blocks, cur, hdr = [], None, None
    for k, l in enumerate(body):
        mm = re.match(r"^\.(LBB\d+_\d+):", l)
        if mm:
            # the INNERMOST loop the block belongs to: "in Loop: Header=BBx_y Depth=d" names it; a block that is itself a loop header
            # ("This (Inner) Loop Header: Depth=d", on the label's line or among the comment lines under it) belongs to its own loop
            hdr = None
            for q in range(k, min(k + 8, len(body))):
                if q > k and not body[q].strip().startswith(";"):
                    break
                h = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", body[q])
                if h:
                    hdr = (mm.group(1)[1:], int(h.group(1)))
                    break
                h = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", body[q])
                if h:
                    hdr = (h.group(1), int(h.group(2)))
                    break
            cur = {"label": mm.group(1), "hdr": hdr, "ins": []}
            blocks.append(cur)
            continue
        t = l.strip()
        if cur is not None and "PM_RARE_PATH" in t:
            cur["ins"].append("@rare")          # the marker PM_RARE_PATH() leaves in the assembly
            continue
        if cur is None or not t or t.startswith((";", ".", "//")):
            continue
        cur["ins"].append(t.split(";")[0].strip())
    # instructions that the stream skips with a wave-level forward branch (the form PM_WAVE_ALL / PM_RARE_PATH compile to) and that
    # carry the PM_RARE_PATH marker are a rare path: split off into blocks of their own, listed, and left out of the common-path
    # totals.  The skipped stretch may span several labelled blocks (an exec-masked region inside the rare path).
    order = {b["label"]: k for k, b in enumerate(blocks)}
    rare_from = {}          # block index -> instruction index from which the block is rare (0: the whole block)
    for k, b in enumerate(blocks):
        for q, ins in enumerate(b["ins"]):
            mm = re.match(r"s_cbranch_(?:scc[01]|vccn?z)\s+\.(LBB\d+_\d+)", ins)
            if not mm or mm.group(1) not in order or order[mm.group(1)] <= k:
                continue
            tgt = order[mm.group(1)]
            skipped = b["ins"][q + 1:] + [i for bb in blocks[k + 1:tgt] for i in bb["ins"]]
            if "@rare" in skipped and tgt - k <= 16:
                rare_from.setdefault(k, q + 1)
                for kk in range(k + 1, tgt):
                    rare_from[kk] = 0
    split = []
    for k, b in enumerate(blocks):
        if k not in rare_from:
            b["rare"] = False
            split.append(b)
        elif rare_from[k] == 0:
            b["rare"] = True
            split.append(b)
        else:
            cut = rare_from[k]
            split.append({"label": b["label"], "hdr": b["hdr"], "ins": b["ins"][:cut], "rare": False})
            split.append({"label": b["label"] + "+", "hdr": b["hdr"], "ins": b["ins"][cut:], "rare": True})
    for b in split:
        b["ins"] = [i for i in b["ins"] if i != "@rare"]
    blocks = split
    # the Runge-Kutta loop: the INNERMOST loop (no child loops) with the most fp64 products — since the tile queue the kernel body sits
    # in a loop of its own, whose straight-line part (re-seeding, remesh, the guards) holds more products than one RK attempt
    parents = set()
    for k, l in enumerate(body):
        mm = re.match(r"^\.(LBB\d+_\d+):", l)
        if mm:
            for q in range(k + 1, min(k + 40, len(body))):
                if not body[q].strip().startswith(";"):
                    break
                if "Child Loop" in body[q]:
                    parents.add(mm.group(1)[1:])
                    break
    prod = Counter()
    for b in blocks:
        if b["hdr"] and b["hdr"][0] not in parents:
            prod[b["hdr"][0]] += sum(bool(re.match(r"v_(fma|fmac|mul)_f64", i)) for i in b["ins"])
    rk = max(prod, key=prod.get)
    tot = Counter()
    per_block = []
    for b in blocks:
        if b["hdr"] and b["hdr"][0] == rk:
            c = Counter(classify(i) for i in b["ins"])
            per_block.append((b["label"] + (" (rare)" if b["rare"] else ""), sum(c.values()), c))
            if not b["rare"]:
                tot.update(c)
    return m.group(1), rk, tot, per_block


def main():
    unit, prefix = sys.argv[1], sys.argv[2]
    name, rk, tot, per_block = budget(unit, prefix)
    valu = sum(v for k, v in tot.items() if k.split(" ")[0] in ("fp64", "cvt", "mov", "select", "lane", "int") or k.startswith("other:v_"))
    print(f"{name}\nRK loop header {rk}, common path (blocks behind a wave-uniform skip left out): {sum(tot.values())} instructions, {valu} VALU")
    for k in list(CLASSES) + sorted(x for x in tot if x.startswith("other:")):
        if tot.get(k):
            print(f"  {k:72s} {tot[k]:5d}")
    if "-v" in sys.argv:
        for lab, n, c in per_block:
            print(f"    {lab:12s} {n:5d}  " + ", ".join(f"{k.split(' (')[0]}={v}" for k, v in c.most_common(6)))


if __name__ == "__main__":
    main()
