"""Inputs for julia/reference_twins.jl: for each of the five native models a few particle rows (x_{t-1}, x_t), the step's data
vector and the ORACLE's deterministic quantities at those values -- log p(y_t | x_t) and, for the rejuvenation moves, the
log-likelihood ratio between two latent values.  On a box with Julia + Gen the twins script evaluates the same
quantities with Gen (Gen.project / Gen.logpdf on @gen twins of the models) and writes tests/golden/ref_twin_weights.txt;
tests/test_reference_twins.py then pins the oracle to the reference's own arithmetic (it skips while that file is absent).

    python tests/golden/make_twin_inputs.py        # rewrites tests/golden/twin_inputs.txt

Line format (whitespace separated; everything a 17-significant-digit decimal, exact for Float64):
    model t n_obs obs... d xprev... xcur... loglik_oracle"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import gpf_amd as g  # noqa: E402  (descriptors only: no GPU needed)
from oracle import oracle as o  # noqa: E402

N, STEPS = 12, 3


def main():
    lines = []
    for name in ("lgssm2", "bearings4", "sv1", "object_motion", "line_model"):
        m = g.models.by_name(name)
        ys = g.models.simulate(m, STEPS + 1) if name != "line_model" else np.array([g.models.line_obs(t, 0.5) for t in range(1, STEPS + 2)])
        f = o.OracleFilter(m.model_id, m.params, N, 3, keep_prev=True).initialize(ys[0])
        for t in range(1, STEPS + 1):
            f.update(ys[t])
            ll = np.empty(N)
            o.lib().o_loglik_rows(m.model_id, f.params, np.ascontiguousarray(f.rows), f.W, N, np.ascontiguousarray(ys[t]), ll)
            for i in range(N):
                cur, prev = f.rows[i, :m.dim], f.rows[i, m.dim:2 * m.dim]
                vals = [*ys[t], m.dim, *prev, *cur, ll[i]]
                lines.append(" ".join([name, str(t + 1), str(len(ys[t]))] + [repr(float(v)) if not isinstance(v, int) else str(v) for v in vals]))
    with open(os.path.join(HERE, "twin_inputs.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print(len(lines), "lines")


if __name__ == "__main__":
    main()
