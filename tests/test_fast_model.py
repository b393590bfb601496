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


# ---- the 127-step loop (twisted-ate form) and the membership tests that guard it --------------------------------------------
def _g2_affine(j):
    X, Y, Z = j
    zi = fm.f2_inv(Z); zi2 = fm.f2_sqr(zi)
    return (fm.f2_mul(X, zi2), fm.f2_mul(Y, fm.f2_mul(zi2, zi)))


def test_short_loop_tate_matches_oracle():
    rng = SplitMix64(24)
    ps = [g1_gen()] + [g1_mul(g1_gen(), rng.below(R)) for _ in range(3)]
    qs = [g2_gen()] + [g2_mul(g2_gen(), rng.below(R)) for _ in range(3)]
    rc, o, _ = pair(3, np.concatenate(ps), np.concatenate(qs))
    assert rc == 0
    want = fq12_from_arr(o)
    for i in range(4):
        got = fm.tate_short(*_pq(ps[i], qs[i]))
        assert got is not None and tuple(fm.to_ref_order(got)) == want[i]
    assert (2 * fm.X2 - 1) % R != 0 and (fm.X2 * fm.X2 - fm.X2 + 1) == R                  # the correction exponent is a unit mod r = x^4 - x^2 + 1


def test_short_loop_g1_membership_is_r_torsion():
    """V = x^2 P == (BETA x, -y)  <=>  r P = infinity, on curve points inside and outside G1 (cofactor parts, small orders)"""
    rng = SplitMix64(25)
    q = _pq(g1_gen(), g2_gen())[1]
    inside = [g1_from_arr(g1_mul(g1_gen(), rng.below(R)))[0] for _ in range(3)]
    outside = [p for _, p in degenerate_g1_points()] + [py_g1_curve_point(s) for s in range(40, 46)]
    for p in inside + outside:
        assert fm.g1_on_curve(p)
        assert fm.miller_short(p, q)[1] == (py_g1_mul(p, R) is None), p
    assert all(fm.miller_short(p, q)[1] for p in inside) and not any(fm.miller_short(p, q)[1] for p in outside[-6:])


def test_short_loop_g2_membership_is_r_torsion():
    """psi(Q) == [x] Q  <=>  r Q = infinity on E'(Fq2); off-curve points are refused"""
    rng = SplitMix64(26)
    for _ in range(3):
        t = py_twist_point(rng)
        assert fm.g2_on_curve(t)
        assert fm.g2_jac_mul(t, R) is not None and not fm.g2_in_subgroup(t)                # a random twist point is outside G2 ...
        g = _g2_affine(fm.g2_jac_mul(t, G2_COFACTOR))
        assert fm.g2_jac_mul(g, R) is None and fm.g2_in_subgroup(g)                        # ... its cofactor multiple is inside
        assert fm.tate_short(g1_from_arr(g1_gen())[0], t) is None
    q = _pq(g1_gen(), g2_mul(g2_gen(), rng.below(R)))[1]
    assert fm.g2_in_subgroup(q)
    off = (q[0], fm.f2_add(q[1], (1, 0)))
    assert not fm.g2_on_curve(off) and not fm.g2_in_subgroup(off)


# ---- decisions through the optimal-ate loop (verification entry points) -------------------------------------------------------------
def _f12_from_ref(t):
    """inverse of fm.to_ref_order for comparing / multiplying oracle values inside the model"""
    probe = tuple(tuple((6 * i + 2 * j, 6 * i + 2 * j + 1) for j in range(3)) for i in range(2))
    order = fm.to_ref_order(probe)
    flat = [None] * 12
    for pos, tag in enumerate(order): flat[tag] = t[pos]
    return tuple(tuple((flat[6 * i + 2 * j], flat[6 * i + 2 * j + 1]) for j in range(3)) for i in range(2))


def test_ate_decisions_equal_tate_decisions():
    """prod tate(P_k, Q_k) == 1 (the faithful oracle's values) <=> prod ate(Q_k, P_k) == 1, on products that are one by construction and on near misses;
    and the ate value is a pairing: a(Q, P)^n = a([n]Q, P) = a(Q, [n]P)."""
    rng = SplitMix64(27)
    a, b, c = (rng.below(R) for _ in range(3))
    P, Qg = g1_gen(), g2_gen()
    # e(aP, bQ) e(-cP, Q) e(P, (c - ab) Q) == 1
    g1s = [g1_mul(P, a), g1_mul(P, R - c), P]
    g2s = [g2_mul(Qg, b), Qg, g2_mul(Qg, (c - a * b) % R)]
    near = [g2s[0], g2s[1], g2_mul(Qg, (c - a * b + 1) % R)]
    rc, o, _ = pair(3, np.concatenate(g1s + g1s), np.concatenate(g2s + near))
    assert rc == 0
    vals = [_f12_from_ref(v) for v in fq12_from_arr(o)]
    prod = lambda vs: fm.f12_mul(fm.f12_mul(vs[0], vs[1]), vs[2])
    assert prod(vals[:3]) == fm.F12_1 and prod(vals[3:]) != fm.F12_1
    ps = [_pq(p, Qg)[0] for p in g1s]
    assert fm.ate_product_is_one(ps, [_pq(P, q)[1] for q in g2s]) is True
    assert fm.ate_product_is_one(ps, [_pq(P, q)[1] for q in near]) is False
    p0, q0 = _pq(P, Qg)
    base = fm.ate_product([p0], [q0])
    assert base != fm.F12_1 and fm.f12_pow(base, R) == fm.F12_1
    assert fm.ate_product([_pq(g1_mul(P, 7), Qg)[0]], [q0]) == fm.f12_pow(base, 7)
    assert fm.ate_product([p0], [_pq(P, g2_mul(Qg, 11))[1]]) == fm.f12_pow(base, 11)


def test_three_times_the_exact_final_exponent():
    """final_exp_3h(f) = final_exp_fast(f)^3 for arbitrary f (the identity 3h = (x-1)^2 (x+q)(x^2+q^2-1) + 3), and cubing is a bijection of G_T (3 is prime to r)"""
    rng = SplitMix64(29)
    f = tuple(tuple((rng.below(Q), rng.below(Q)) for _ in range(3)) for _ in range(2))
    e = fm.final_exp_fast(f)
    assert fm.final_exp_3h(f) == fm.f12_mul(fm.f12_sqr(e), e)
    assert R % 3 != 0 and fm.final_exp_3h(fm.F12_1) == fm.F12_1


def test_ate_route_refuses_arguments_outside_their_groups():
    """points outside G1 / G2 or off their curves never reach the ate loop (they keep the 255-step loop / the reference's chain): the model returns None,
    and the G2 test the loop gets for free agrees with r Q = infinity"""
    rng = SplitMix64(28)
    p0, q0 = _pq(g1_gen(), g2_gen())
    for _, p in degenerate_g1_points():
        assert fm.ate_product_is_one([p], [q0]) is None
    assert fm.ate_product_is_one([py_g1_curve_point(41)], [q0]) is None
    assert fm.ate_product_is_one([(p0[0], (p0[1] + 1) % Q)], [q0]) is None
    t = py_twist_point(rng)
    assert fm.ate_line_table(t)[1] is False and fm.ate_product_is_one([p0], [t]) is None
    g = _g2_affine(fm.g2_jac_mul(t, G2_COFACTOR))
    assert fm.ate_line_table(g)[1] is True and fm.ate_line_table(g)[1] == fm.g2_in_subgroup(g)
    assert fm.ate_product_is_one([p0], [(q0[0], fm.f2_add(q0[1], (1, 0)))]) is None
    assert len(fm.ate_line_table(q0)[0]) == fm.ATE_LINES == 68
