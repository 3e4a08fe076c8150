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
