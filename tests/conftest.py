import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_soak: GPU tests beyond one representative per code path (the wide multi-process matrices); they carry `gpu` too but "
                                       "only run when the -m expression names gpu_soak (tools/gpu.sh soak) -- the driver's `-m gpu` stays inside its time limit")


def soak_grid(*axes, keep):
    """the full product of the axes as pytest params; the combinations `keep(*combo)` rejects carry the gpu_soak marker"""
    import itertools
    return [pytest.param(*c, marks=() if keep(*c) else (pytest.mark.gpu_soak,)) for c in itertools.product(*axes)]


def pytest_collection_modifyitems(config, items):
    """gpu_soak items are deselected unless the marker expression asks for them by name"""
    if "gpu_soak" in (config.getoption("markexpr", "") or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("gpu_soak") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def built():
    """Build (if stale) the HIP library and the oracle once per session."""
    import __graft_entry__ as ge
    ge.build()
    return True


@pytest.fixture(scope="session")
def g(built):
    import gpf_amd
    return gpf_amd


@pytest.fixture(scope="session")
def o(built):
    from oracle import oracle
    oracle.lib()
    return oracle
