"""The C-ABI library loads without a GPU and exports every symbol include/gpf.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "gpf.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gpf_[A-Za-z_0-9]+)\s*\(", txt)))


def test_header_symbols_exported(g):
    L = ctypes.CDLL(g._lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/gpf.h but not exported"
    assert sorted(s[0] for s in g._lib.SYMBOLS) == names, "ctypes table out of sync with include/gpf.h"
    assert L.gpf_abi_version() == 1


def test_no_cpu_fallback(g):
    """Without a GPU the product must fail loudly, not fall back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(g.ErrorException, match="no HIP device|no CPU fallback|hip"):
        g.pf_initialize(g.models.lgssm2(), (1,), [0.0, 0.0], 16)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under the product package may import, include, link or load it."""
    pkg = os.path.join(ROOT, "genparticlefilters.jl_amd")
    bad = re.compile(r"(^\s*(from|import)\s+oracle\b|#include\s*[\"<][^\">]*oracle|liboracle|oracle/_build|oracle\.oracle)", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".jl")):
                src = open(os.path.join(dirpath, f)).read()
                assert not bad.search(src), f"{f} uses oracle/"


def test_julia_glue_matches_header():
    """julia/GenParticleFiltersAMD.jl cannot be run here (no Julia in the image); at least every ccall it makes must name an
    entry point of include/gpf.h with the same number of arguments"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    jl = open(os.path.join(root, "julia", "GenParticleFiltersAMD.jl")).read()
    hdr = open(os.path.join(root, "include", "gpf.h")).read()
    names = set(re.findall(r"\(:([a-z_A-Z0-9]+), libgpf\)", jl))
    assert len(names) >= 20
    for n in sorted(names):
        h = re.search(r"\b%s\s*\(([^;]*?)\);" % n, hdr, re.S)
        assert h, f"{n} is not declared in include/gpf.h"
        m = re.search(r"\(:%s, libgpf\), [A-Za-z]+, \(([^)]*)\)" % n, jl)
        assert m, n
        ja = [a for a in m.group(1).split(",") if a.strip()]
        ha = [a for a in h.group(1).split(",") if a.strip() and a.strip() != "void"]
        assert len(ja) == len(ha), f"{n}: {len(ja)} ccall argument types, {len(ha)} parameters in the header"


# what a ccall argument type may be for a C parameter type of include/gpf.h (Julia's C interface: Cint = Int32, Clonglong = Int64, Cdouble = Float64;
# Ref{T} and Ptr{T} both pass a T*; a gpf_handle is an opaque pointer)
_JULIA_FOR_C = {
    "gpf_handle": {"Ptr{Cvoid}"},
    "gpf_handle*": {"Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"},
    "gpf_config*": {"Ref{GpfConfig}", "Ptr{GpfConfig}"},
    "int32_t": {"Cint", "Int32"}, "int": {"Cint", "Int32"},
    "int64_t": {"Int64", "Clonglong"},
    "uint32_t": {"UInt32", "Cuint"}, "uint64_t": {"UInt64", "Culonglong"},
    "double": {"Cdouble", "Float64"},
    "double*": {"Ptr{Cdouble}", "Ref{Cdouble}", "Ptr{Float64}", "Ref{Float64}"},
    "int32_t*": {"Ptr{Cint}", "Ref{Cint}", "Ptr{Int32}", "Ref{Int32}"},
    "int64_t*": {"Ptr{Int64}", "Ref{Int64}", "Ptr{Clonglong}"},
    "uint64_t*": {"Ptr{UInt64}", "Ref{UInt64}"},
    "void*": {"Ptr{Cvoid}", "Ptr{UInt8}"},
}
_JULIA_RETURN = {"gpf_status": {"Cint"}, "int": {"Cint"}, "char*": {"Cstring", "Ptr{UInt8}"}, "void": {"Cvoid"}, "int32_t": {"Cint", "Int32"},
                 "uint64_t": {"UInt64"}, "double": {"Cdouble", "Float64"}}


def _c_prototypes():
    """name -> (return type, [parameter types]) of every function include/gpf.h declares; `const` dropped, pointers glued to the type"""
    import re
    hdr = open(os.path.join(ROOT, "include", "gpf.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z_0-9 ]*?[ *]+)\b(gpf_[A-Za-z_0-9]+)\s*\(([^;{}]*?)\)\s*;", hdr):
        ret = m.group(1).replace("const", "").strip().replace(" *", "*").replace(" ", "")
        params = []
        for a in [x.strip() for x in m.group(3).split(",")]:
            if not a or a == "void":
                continue
            a = re.sub(r"\bconst\b", "", a).strip()
            t = re.sub(r"\s*[A-Za-z_][A-Za-z_0-9]*$", "", a) if re.search(r"[ *][A-Za-z_][A-Za-z_0-9]*$", a) else a      # drop the parameter's name
            params.append(t.replace(" ", ""))
        protos[m.group(2)] = (ret, params)
    return protos


def _split_types(tup):
    """'Ptr{Cvoid}, Ref{Ptr{Cvoid}}' -> the top-level comma-separated types"""
    out, depth, cur = [], 0, ""
    for ch in tup:
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def test_julia_glue_argument_types_match_header():
    """The glue has never run (no Julia in the image), and an arity check passes a `Cint` where the header wants an `int64_t`: map EVERY ccall's return and
    argument types onto the C prototype (Cint <-> int32_t, Int64 / Clonglong <-> int64_t, Cdouble <-> double, Ptr{T} / Ref{T} <-> T*, UInt64 <-> uint64_t,
    Ptr{Cvoid} <-> gpf_handle / void*); the calls through a symbol variable (`_scalar`, `_vector`) against every symbol they are given; and the field
    order and types of `struct GpfConfig` against the header's gpf_config."""
    import re
    jl = open(os.path.join(ROOT, "julia", "GenParticleFiltersAMD.jl")).read()
    protos = _c_prototypes()
    assert len(protos) >= 90 and protos["gpf_resize"] == ("gpf_status", ["gpf_handle", "int64_t", "int32_t", "double", "int32_t", "int32_t*"])

    def check(name, ret, types, where):
        assert name in protos, f"{where}: {name} is not declared in include/gpf.h"
        cret, cparams = protos[name]
        assert ret in _JULIA_RETURN[cret], f"{where}: {name} returns {cret}, the ccall says {ret}"
        assert len(types) == len(cparams), f"{where}: {name} takes {len(cparams)} arguments, the ccall passes {len(types)}"
        for k, (jt, ct) in enumerate(zip(types, cparams)):
            assert jt in _JULIA_FOR_C[ct], f"{where}: argument {k + 1} of {name} is `{ct}`, the ccall passes `{jt}`"

    n_checked = 0
    for m in re.finditer(r"ccall\(\(:([a-z_A-Z0-9]+), libgpf\), ([A-Za-z]+), \(", jl):
        # the argument-type tuple: up to its matching parenthesis
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(jl[i], 0); i += 1
        check(m.group(1), m.group(2), _split_types(jl[m.end():i - 1]), f"line {jl.count(chr(10), 0, m.start()) + 1}")
        n_checked += 1
    assert n_checked >= 60
    # the two helpers that take the entry point as a symbol
    helpers = {"_scalar": ["Ptr{Cvoid}", "Ref{Cdouble}"], "_vector": ["Ptr{Cvoid}", "Ptr{Cdouble}", "Int64"]}
    for hname, types in helpers.items():
        body = re.search(r"function %s\(s, sym\)(.*?)^end" % hname, jl, re.S | re.M).group(1)
        assert "(%s)" % ", ".join(types) in body, f"{hname}: its ccall no longer passes {types}"
        syms = re.findall(r"%s\(s, :([a-z_A-Z0-9]+)\)" % hname, jl)
        assert syms, hname
        for sym in syms:
            check(sym, "Cint", types, hname)
    # struct GpfConfig: same fields, same order, matching types
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "gpf.h")).read(), flags=re.S)
    cfields = re.findall(r"^\s*(?:const\s+)?([A-Za-z_0-9]+\s*\*?)\s*([a-z_]+);", re.search(r"typedef struct \{(.*?)\} gpf_config;", hdr, re.S).group(1), re.M)
    jfields = re.findall(r"([a-z_]+)::([A-Za-z0-9{}]+)", re.search(r"struct GpfConfig\n(.*?)\nend", jl, re.S).group(1))
    assert [f for _, f in cfields] == [f for f, _ in jfields], "GpfConfig: field names / order differ from gpf_config"
    for (ct, name), (_, jt) in zip(cfields, jfields):
        assert jt in _JULIA_FOR_C[ct.replace(" ", "")], f"GpfConfig.{name}: `{ct.strip()}` in the header, `{jt}` in the glue"


def test_julia_glue_has_no_shadowed_status_helper():
    """The resamplers take a keyword called `check` (src/resample.jl:43-46).  A status helper of the same name is shadowed inside those
    methods (`:warn(s, st)` -> MethodError after the ccall has already mutated the state): the helper is `_status`, no bare `check(`
    call may remain, and every method with a `check` keyword reports its status through `_status`."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    jl = open(os.path.join(root, "julia", "GenParticleFiltersAMD.jl")).read()
    assert not re.search(r"(?<![\w.:])check\(", jl), "a call of `check(...)`: shadowed wherever a `check=` keyword is in scope"
    assert "_status(state, st) = st == 0" in jl
    # every `function ...; check=...)` body that makes a ccall checks its status with _status
    for m in re.finditer(r"^function (\w+!?)\(([^\n]*(?:\n[^\n]*)?check=[^\n]*)\n(.*?)^end", jl, re.S | re.M):
        body = m.group(3)
        if "ccall" in body:
            assert "_status(" in body, m.group(1)
