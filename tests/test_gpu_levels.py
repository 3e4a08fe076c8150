"""The levels the weight scan leaves for k_search_fine (4-byte keys per 1024 cells, 16-bit offsets per 16-cell group,
8-bit offsets per cell), read back through gpf_debug_levels and recomputed with NumPy from the scan's own CDF -- the producer side
of the search on its own, so that a wrong level cannot hide behind the exact tie-breaks of the consumer."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _level(g, st, which, dtype, cap):
    L = g._lib.load()
    buf = np.zeros(cap, dtype); nb = C.c_int64(buf.nbytes)
    assert L.gpf_debug_levels(st._h, which, buf.ctypes.data_as(C.c_void_p), C.byref(nb)) == 0, L.gpf_last_error(st._h)
    return buf[: nb.value // buf.itemsize]


def _bitlen(a):
    return np.array([int(x).bit_length() for x in a], dtype=np.int64)


@pytest.mark.parametrize("N,weights", [(100_000, "filter"), (1_000_000, "filter"), (300_001, "collapsed"), (65_536, "equal")])
def test_fine_levels_match_numpy(g, o, N, weights):
    model = g.models.lgssm2(); ys = g.models.simulate(model, 3)
    st = g.pf_initialize(model, (1,), ys[0], N, seed=4)
    if weights == "collapsed":
        st.log_weights = np.where(np.arange(N) % 9973 == 5, 0.0, -35.0 - 1e-4 * np.arange(N))
    elif weights == "equal":
        st.log_weights = np.zeros(N)                                   # S = 2^62 exactly: the saturated prefix
    g.pf_resample(st, "multinomial", check=False)
    nt = (N + 2047) // 2048
    cdf = _level(g, st, 0, np.uint64, nt * 2048)
    k1 = _level(g, st, 6, np.uint32, nt * 2); d16 = _level(g, st, 7, np.uint16, nt * 128); o8 = _level(g, st, 8, np.uint8, nt * 2048)
    assert k1.size == nt * 2 and d16.size == nt * 128 and o8.size == nt * 2048, "the scan did not write k_search_fine's levels"
    sat = np.minimum(cdf, np.uint64((1 << 62) - 1))
    ends = sat[1023::1024]
    assert np.array_equal(k1, (ends >> np.uint64(30)).astype(np.uint32))
    klo = np.concatenate([[0], k1[:-1]]).astype(np.uint64); khi = k1.astype(np.uint64)
    w = (khi - klo + 1).astype(object)
    sh = np.array([14 + (0 if x <= 1 else int(x - 1).bit_length()) for x in w], dtype=np.uint64)
    kb = klo << np.uint64(30)
    gend = sat[15::16].reshape(-1, 64)                                 # prefix at the end of every 16-cell group, per super-group
    dexp = ((gend - kb[:, None]) >> sh[:, None])
    assert dexp.max() < 65536 and np.array_equal(d16.reshape(-1, 64), dexp.astype(np.uint16))
    dprev = np.concatenate([np.zeros((dexp.shape[0], 1), np.uint64), dexp[:, :-1]], axis=1)
    dd = (dexp - dprev).reshape(-1)
    bits = _bitlen(dd)
    sh8 = np.repeat(sh, 64) + np.maximum(bits - 8, 0).astype(np.uint64)
    base = np.repeat(kb, 64) + (dprev.reshape(-1) << np.repeat(sh, 64))
    oexp = (sat.reshape(-1, 16) - base[:, None]) >> sh8[:, None]
    assert oexp.max() <= 255 and np.array_equal(o8.reshape(-1, 16), oexp.astype(np.uint8))
    st.close()
