"""Run one sequence of tests/test_gpu_fuzz.py::test_random_api_sequences and print its full operation log on failure:
   GPF_FUZZ_OFFSET=... python tools/dbg_fuzz.py SEED"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPF_FUZZ_SEEDS", "1")
import gpf_amd as g
from oracle import oracle as o
import test_gpu_fuzz as T
seed = int(sys.argv[1])
try:
    T.test_random_api_sequences(g, o, seed)
    print("passed")
except BaseException as e:                                           # noqa: BLE001
    tb = e.__traceback__
    while tb is not None:
        if tb.tb_frame.f_code.co_name == "test_random_api_sequences":
            loc = tb.tb_frame.f_locals
            print("N", loc.get("N"), "model", loc.get("name"), "n now", loc["st"].n_particles, "op", loc.get("op"))
            for i, x in enumerate(loc.get("log", [])):
                print(i, x)
            import numpy as np
            st, orc = loc["st"], loc["orc"]
            lw = st.log_weights
            print("device lw: nan", int(np.isnan(lw).sum()), "-inf", int(np.isneginf(lw).sum()), "+inf", int(np.isposinf(lw).sum()), "finite", int(np.isfinite(lw).sum()),
                  "equal to oracle", np.array_equal(lw, orc.lw, equal_nan=True))
            nw = g.get_norm_weights(st); print("norm weights (as is):", nw[:4], "ess", g.get_ess(st), "lml", g.get_lml_est(st), "oracle lml", orc.log_ml_estimate())
            st.log_weights = lw                                            # (drops every cached summary)
            nw = g.get_norm_weights(st); print("norm weights after re-setting the weights:", nw[:4], "ess", g.get_ess(st))
        tb = tb.tb_next
    print("".join(traceback.format_exception_only(type(e), e))[:1500])
