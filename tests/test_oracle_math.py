"""Pins the oracle's deterministic math: Philox4x32-10 known-answer vectors (Random123 kat_vectors),
accuracy of exp/log/sincos/atan2 against libm, and fixed-point weight edge cases."""
import math

import numpy as np


def test_philox_known_answers(o):
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        out = np.zeros(4, np.uint32)
        o.lib().o_philox_d(*ctr, *key, out)
        assert tuple(int(x) for x in out) == want


def _ulps(got, want):
    want = np.asarray(want, np.float64)
    return np.abs(got - want) / np.spacing(np.abs(want))


def test_math_accuracy_vs_libm(o):
    rng = np.random.default_rng(0)
    n = 200_000
    L = o.lib()
    out, out2 = np.empty(n), np.empty(n)
    x = rng.uniform(-700, 20, n)
    L.o_math_vec(0, x, x, n, out, out2)
    assert _ulps(out, np.exp(x)).max() < 2.0
    y = np.exp(rng.uniform(-700, 700, n))
    L.o_math_vec(1, y, y, n, out, out2)
    assert _ulps(out, np.log(y)).max() < 2.0
    u = rng.uniform(0, 1, n)
    L.o_math_vec(2, u, u, n, out, out2)
    assert np.abs(out - np.sin(2 * np.pi * u)).max() < 1e-15
    assert np.abs(out2 - np.cos(2 * np.pi * u)).max() < 1e-15
    a, b = rng.uniform(-4, 4, n), rng.uniform(-4, 4, n)
    L.o_math_vec(3, a, b, n, out, out2)
    assert _ulps(out, np.arctan2(a, b)).max() < 4.0
    assert L.o_atan2_d(0.0, -1.0) == math.pi and L.o_atan2_d(0.0, 0.0) == 0.0


def test_normals_moments(o):
    z = o.normals(12345, 0, 2_000_000)
    assert abs(z.mean()) < 4e-3 and abs(z.var() - 1.0) < 5e-3
    assert abs((z ** 3).mean()) < 1e-2 and abs((z ** 4).mean() - 3.0) < 3e-2
    u = np.array([o.lib().o_u52_d(7, i, 0, 0, 1) for i in range(20000)])
    assert 0.0 < u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 1e-2


def test_fixed_point_weights(o):
    L = o.lib()
    for n, K in [(1, 52), (100, 52), (2048, 51), (10**6, 42), (8 * 10**6, 39), (2**31 - 1, 31)]:
        assert L.o_fix_K(n) == K and n * 2**K <= 2**62
    assert L.o_exp_fix_d(0.0, 42) == 2**42
    assert L.o_exp_fix_d(-np.inf, 42) == 0 and L.o_exp_fix_d(-800.0, 42) == 0
    assert L.o_exp_fix_d(-math.log(2.0), 42) in (2**41 - 1, 2**41, 2**41 + 1)
    # weights below half a quantum vanish, the maximum never does
    assert L.o_exp_fix_d(-43 * math.log(2.0), 42) in (0, 1)


def test_weighted_sums_follow_the_binary_tree(g, o):
    """mean / var (src/statistics.jl:13-14, 48-50): the reference adds the terms one after the other; the spec adds the SAME terms
    by the perfect binary tree over their indices in chunks of 2048.  Checked against a plain-Python restatement of
    that tree (bit for bit) and against math.fsum / the sequential sum (to rounding error)."""
    import math

    def tree(t):
        t = list(t)
        while True:
            out = []
            for b in range(0, len(t), 2048):
                buf = t[b:b + 2048] + [0.0] * (2048 - len(t[b:b + 2048]))
                w = 1
                while w < 2048:
                    for i in range(0, 2048, 2 * w):
                        buf[i] = buf[i] + buf[i + w]
                    w *= 2
                out.append(buf[0])
            if len(out) == 1:
                return out[0]
            t = out

    m = g.models.lgssm2()
    for n in (1, 5, 2048, 2049, 5000):
        f = o.OracleFilter(m.model_id, m.params, n, 3).initialize(g.models.simulate(m, 1)[0])
        f.lw = -20.0 * np.random.default_rng(n).random(n)
        s = f.summary()
        terms = [(float(q) / float(s.S)) * float(v) for q, v in zip(s.q, f.rows[:, 0])]
        assert f.mean(0) == tree(terms)
        assert abs(f.mean(0) - math.fsum(terms)) <= 1e-13 * max(1.0, sum(abs(x) for x in terms))
        seq = 0.0
        for x in terms:
            seq += x
        assert abs(f.mean(0) - seq) <= 1e-12 * max(1.0, sum(abs(x) for x in terms))       # the reference's order, to rounding error
        mu = f.mean(0)
        assert f.var(0) == tree([(float(q) / float(s.S)) * ((float(v) - mu) * (float(v) - mu)) for q, v in zip(s.q, f.rows[:, 0])])
