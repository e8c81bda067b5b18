"""The C-ABI shared library loads without a GPU and exports every symbol that
include/markovmodels_amd.h declares (no compute calls here)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "markovmodels_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(mm):
    syms = declared_symbols()
    assert len(syms) >= 14
    lib = C.CDLL(mm.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
    assert sorted(mm.SYMBOLS) == syms, "the ctypes binding must bind exactly the header's entry points"
    lib.mm_abi_version.restype = C.c_int
    assert lib.mm_abi_version() == 4


def test_no_torch_types_in_the_abi():
    text = open(os.path.join(ROOT, "include", "markovmodels_amd.h")).read()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert "torch" not in code.lower() and "at::" not in code and "std::" not in code
    assert 'extern "C"' in text


def test_product_does_not_import_the_oracle():
    """The product package must never route through oracle/ (no CPU fallback)."""
    pkg = os.path.join(ROOT, "markovmodels.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "mm_oracle" not in src and "libmm_oracle" not in src, f


# ---- the Julia shim (julia/MarkovModelsAMD.jl) cannot be parsed by a Julia here: its ccall tuples are checked statically ----
def _split_top(s):
    """split at top-level commas (parentheses, braces and brackets nest; string literals are skipped)"""
    out, depth, cur, i = [], 0, [], 0
    while i < len(s):
        ch = s[i]
        if ch == '"':
            j = i + 1
            while s[j] != '"' or s[j - 1] == "\\":
                j += 1
            cur.append(s[i : j + 1])
            i = j + 1
            continue
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
        i += 1
    if "".join(cur).strip():
        out.append("".join(cur).strip())
    return out


def _balanced(text, start):
    """text[start] == '(' -> index just past its matching ')'"""
    depth, i = 0, start
    while True:
        ch = text[i]
        if ch == '"':
            i += 1
            while text[i] != '"' or text[i - 1] == "\\":
                i += 1
        elif ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1


def c_prototypes():
    """name -> (return class, [argument classes]) from the header; classes: i32, i64, f32, f64, ptr, size"""
    text = open(os.path.join(ROOT, "include", "markovmodels_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)

    def cls(t):
        t = t.strip()
        if "*" in t or "[" in t or re.search(r"\bmm_(fsm|batch|statemap)_t\b", t):
            return "ptr"
        t = re.sub(r"\bconst\b", "", t).split()
        base = " ".join(t[:-1]) if len(t) > 1 else t[0]  # (drop the parameter name)
        return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "float": "f32", "double": "f64", "size_t": "size", "void": "void"}[base]

    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(mm_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        rc = "ptr" if "*" in ret else {"int": "i32", "int64_t": "i64", "size_t": "size"}[re.sub(r"\bconst\b", "", ret).strip()]
        protos[name] = (rc, [] if args in ("", "void") else [cls(a) for a in _split_top(args)])
    return protos


def julia_class(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t == "Cstring":
        return "ptr"
    return {"Cint": "i32", "Int32": "i32", "Int64": "i64", "Cfloat": "f32", "Float32": "f32", "Cdouble": "f64", "Float64": "f64",
            "Csize_t": "size", "Cvoid": "void"}[t]


def test_julia_shim_ccalls_match_the_header():
    """Every `ccall((:mm_x, LIB), ret, (types...), args...)` of the Julia shim against the prototype of mm_x in the header: the
    number of argument types, each type's class and width (Cint / Int64 / Cfloat / Cdouble / pointer), the return type, and as many
    values as types.  The shim must also check mm_abi_version against the header's MM_ABI_VERSION when it is loaded."""
    protos = c_prototypes()
    src = open(os.path.join(ROOT, "julia", "MarkovModelsAMD.jl")).read()
    seen, n = set(), 0
    for m in re.finditer(r"ccall\(", src):
        end = _balanced(src, m.end() - 1)
        parts = _split_top(src[m.end() : end - 1])
        target, ret, types, values = parts[0], parts[1], parts[2], parts[3:]
        tm = re.match(r"\(\s*:?(\w+)\s*,\s*LIB\s*\)", target)
        assert tm, target
        names = [tm.group(1)] if tm.group(1).startswith("mm_") else ["mm_alpharecursion_f32", "mm_betarecursion_f32", "mm_maxstateposteriors_f32"]
        assert tm.group(1).startswith("mm_") or tm.group(1) == "sym", target  # (`sym`: the three recursion entries share one signature)
        assert types.startswith("(") and types.endswith(")"), types
        jt = [julia_class(t) for t in _split_top(types[1:-1])]
        for name in names:
            assert name in protos, f"{name} is not declared in the header"
            rc, ct = protos[name]
            assert julia_class(ret) == rc, (name, ret, rc)
            assert jt == ct, f"{name}: Julia passes {jt}, the header declares {ct}"
            assert len(values) == len(ct), f"{name}: {len(values)} values for {len(ct)} argument types"
            seen.add(name)
        n += 1
    assert n >= 30 and "mm_abi_version" in seen and "mm_batch_set_mark_policy" in seen and "mm_batch_set_gamma_mode" in seen
    hv = int(re.search(r"#define MM_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "markovmodels_amd.h")).read()).group(1))
    assert re.search(rf"const MM_ABI_VERSION = {hv}\b", src), "the shim's MM_ABI_VERSION must be the header's"
    assert "function __init__()" in src and "mm_abi_version" in src[src.index("function __init__()") :][:400]
