"""The fast pairing algorithm used by the HIP kernels (python-int model in
oracle/fast_model.py) must be bit-identical to the faithful oracle's tate()."""
import os, sys
import numpy as np
from zkt_testlib import *
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fast_model as fm
from test_oracle_kats import g1_gen, g2_gen, g1_mul, g2_mul, O
from test_oracle_pairing import pair


def _pq(p_arr, q_arr):
    p = g1_from_arr(p_arr)[0]; (x1, x0), (y1, y0) = g2_from_arr(q_arr)[0]
    return p, ((x0, x1), (y0, y1))


def test_frobenius_constants():
    rng = SplitMix64(21)
    a = tuple(tuple((rng.below(Q), rng.below(Q)) for _ in range(3)) for _ in range(2))
    assert fm.f12_frob(a, 1) == fm.f12_pow(a, Q)
    assert fm.f12_frob(a, 2) == fm.f12_pow(a, Q * Q)


def test_fast_tate_matches_oracle():
    rng = SplitMix64(22)
    ps = [g1_gen()] + [g1_mul(g1_gen(), rng.below(R)) for _ in range(2)]
    qs = [g2_gen()] + [g2_mul(g2_gen(), rng.below(R)) for _ in range(2)]
    rc, o, _ = pair(3, np.concatenate(ps), np.concatenate(qs))
    assert rc == 0
    want = fq12_from_arr(o)
    for i in range(3):
        got = fm.to_ref_order(fm.tate_fast(*_pq(ps[i], qs[i])))
        assert tuple(got) == want[i]


def test_exact_miller_values_and_weil_match_oracle():
    """raw calc_g1_g2 / calc_g2_g1 (pairing.rs:54-55) and weil (pairing.rs:75-84), bit for bit"""
    rng = SplitMix64(23)
    ps = [g1_gen(), g1_mul(g1_gen(), rng.below(R))]
    qs = [g2_gen(), g2_mul(g2_gen(), rng.below(R))]
    P, Qa = np.concatenate(ps), np.concatenate(qs)
    for which, fn in ((0, lambda p, q: fm.calc_g1_g2_exact(p, q)), (1, lambda p, q: fm.calc_g2_g1_exact(q, p)), (2, fm.weil_exact)):
        rc, o, _ = pair(which, P, Qa)
        assert rc == 0
        want = fq12_from_arr(o)
        for i in range(2):
            assert tuple(fm.to_ref_order(fn(*_pq(ps[i], qs[i])))) == want[i], (which, i)
