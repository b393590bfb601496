"""Pin the CPU oracle on every known-answer value the reference's own tests hold
for the hot path.  CPU only."""
import ctypes
import numpy as np
import pytest
from zkt_testlib import *

K = kats()
O = oracle()


def dyn(op, mod, a, b=None):
    m = np.array(int_to_limbs(mod, 6), dtype=np.uint64)
    n = max(1, (mod.bit_length() + 63) // 64)
    assert O.zkto_dyn_set_modulus(ptr(m), n) == 0
    aa = np.array(int_to_limbs(a, 6), dtype=np.uint64)
    bb = np.array(int_to_limbs(b, 6), dtype=np.uint64) if b is not None else None
    o = np.zeros(6, dtype=np.uint64)
    rc = O.zkto_dyn_op(op, ptr(aa), ptr(bb), ptr(o))
    return rc, limbs_to_int(o)


def test_fq_mul_large_number():          # prime_field_elem.rs:600-617
    k = K["fq_mul_large"]
    rc, v = dyn(2, int(k["order"]), int(k["lhs"]), int(k["rhs"]))
    assert rc == 0 and v == int(k["exp"])


def test_inv_small_primes():             # prime_field_elem.rs:625-800
    cases = K["inv_small_primes"]["cases"]
    assert len(cases) == 158
    for order, n, exp in cases:
        rc, v = dyn(3, order, n)
        assert rc == 0 and v == exp, (order, n)


def test_inv_secp256k1():                # prime_field_elem.rs:811-821
    k = K["inv_secp256k1"]
    rc, v = dyn(3, int(k["p_hex"], 16), int(k["a"]))
    assert rc == 0 and v == int(k["exp"])


def test_inv_zero_is_error():            # prime_field_elem.rs:379-382
    rc, _ = dyn(3, 97, 0)
    assert rc == ZKT_ERR_INV_ZERO


def test_pow_various():                  # prime_field_elem.rs:888-909 (modulus 10^8, not prime)
    k = K["pow_various"]
    for base, e, exp in k["cases"]:
        rc, v = dyn(6, int(k["order"]), base, e)
        assert rc == 0 and v == exp


def test_small_field_semantics():        # prime_field_elem.rs:465-600 (new reduces, a-b wraps, neg(0)=0)
    assert dyn(5, 11, 13)[1] == 2
    assert dyn(0, 11, 9, 4)[1] == 2
    assert dyn(1, 11, 3, 5)[1] == 9
    assert dyn(4, 11, 5)[1] == 6 and dyn(4, 11, 0)[1] == 0
    assert dyn(2, 11, 2, 5)[1] == 10


def test_field_ops_vs_python_ints():
    rng = SplitMix64(11)
    for mod, w, pre in ((Q, 6, "fq"), (R, 4, "fr")):
        xs = [rng.below(mod) for _ in range(200)] + [0, 1, mod - 1]
        ys = [rng.below(mod) for _ in range(200)] + [mod - 1, 0, mod - 1]
        a, b = ints_to_arr(xs, w), ints_to_arr(ys, w)
        o = np.zeros_like(a)
        for name, f in (("add", lambda x, y: (x + y) % mod), ("sub", lambda x, y: (x - y) % mod), ("mul", lambda x, y: x * y % mod)):
            assert getattr(O, f"zkto_{pre}_{name}_batch")(ptr(a), ptr(b), ptr(o), len(xs)) == 0
            assert arr_to_ints(o) == [f(x, y) for x, y in zip(xs, ys)]
        nz = ints_to_arr([x for x in xs if x], w)
        o = np.zeros_like(nz)
        assert getattr(O, f"zkto_{pre}_inv_batch")(ptr(nz), ptr(o), len(nz), None) == 0
        assert arr_to_ints(o) == [pow(x, -1, mod) for x in arr_to_ints(nz)]
        idx = ctypes.c_size_t(99)
        assert getattr(O, f"zkto_{pre}_inv_batch")(ptr(a), ptr(np.zeros_like(a)), len(xs), ctypes.byref(idx)) == ZKT_ERR_INV_ZERO
        assert idx.value == 200


# ---- tower: fq_test_helper.rs:9-34 ------------------------------------------
a1, b1, c1, d1 = Q - 3, Q - 5, Q - 7, Q - 9
a2, b2, c2, d2 = (a1, b1), (b1, c1), (c1, d1), (d1, a1)
a6, b6, c6, d6 = a2 + b2 + c2, b2 + c2 + d2, c2 + d2 + a2, d2 + a2 + b2


def tower_op(fn, w, op, x, y=None):
    a = ints_to_arr(list(x), 6).reshape(1, w)
    b = ints_to_arr(list(y), 6).reshape(1, w) if y is not None else None
    o = np.zeros((1, w), dtype=np.uint64)
    assert fn(op, ptr(a), ptr(b), ptr(o), 1) == 0
    return [str(v) for v in arr_to_ints(o.reshape(-1, 6))]


def test_fq2_kats():                     # fq2.rs:166-226
    k = K["fq2"]; x, y = (a1, b1), (c1, d1); f = O.zkto_fq2_op
    assert tower_op(f, 12, 0, x, y) == k["add"]
    assert tower_op(f, 12, 1, x, y) == k["sub"]
    assert tower_op(f, 12, 2, x, y) == k["mul"]
    assert tower_op(f, 12, 3, x) == k["inv_a"]
    assert tower_op(f, 12, 3, y) == k["inv_b"]
    m = [int(v) for v in tower_op(f, 12, 2, x, y)]
    assert tower_op(f, 12, 5, m) == k["reduce_mul"]


def test_fq6_kats():                     # fq6.rs:190-275
    k = K["fq6"]; f = O.zkto_fq6_op
    assert tower_op(f, 36, 0, a6, b6) == k["add"]
    assert tower_op(f, 36, 1, a6, b6) == k["sub"]
    assert tower_op(f, 36, 2, a6, b6) == k["mul"]
    assert tower_op(f, 36, 3, a6) == k["inv_a"]
    assert tower_op(f, 36, 3, b6) == k["inv_b"]
    m = [int(v) for v in tower_op(f, 36, 2, a6, b6)]
    assert tower_op(f, 36, 5, m) == k["reduce_mul"]


def test_fq12_kats():                    # fq12.rs:198-329
    k = K["fq12"]; f = O.zkto_fq12_op
    a12, b12 = a6 + b6, c6 + d6
    assert tower_op(f, 72, 0, a12, b12) == k["add"]
    assert tower_op(f, 72, 1, a12, b12) == k["sub"]
    assert tower_op(f, 72, 2, a12, b12) == k["mul"]
    assert tower_op(f, 72, 3, a12) == k["inv_a"]
    assert tower_op(f, 72, 3, b12) == k["inv_b"]
    # test_pow: Fq12::from(3)^4 == Fq12::from(81)
    three = ints_to_arr([0] * 11 + [3], 6).reshape(1, 72); o = np.zeros((1, 72), dtype=np.uint64)
    e = np.array([4], dtype=np.uint32)
    assert O.zkto_fq12_pow(ptr(three), e.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 1, ptr(o)) == 0
    assert arr_to_ints(o.reshape(-1, 6)) == [0] * 11 + [81]


# ---- G1 -----------------------------------------------------------------------
def g1_gen():
    g = np.zeros((1, G1W), dtype=np.uint64); O.zkto_g1_generator(ptr(g)); return g


def g1_mul(p, k, limbs=4):
    s = ints_to_arr([k], limbs); o = np.zeros((1, G1W), dtype=np.uint64)
    assert O.zkto_g1_mul_batch(ptr(p), ptr(s), limbs, ptr(o), 1, 1) == 0
    return o


def g1_add(p, q):
    o = np.zeros((1, G1W), dtype=np.uint64); assert O.zkto_g1_add_batch(ptr(p), ptr(q), ptr(o), 1) == 0; return o


def test_g1_generator_and_double():      # g1_point.rs:38-47, 224-237
    g = g1_gen()
    assert g1_from_arr(g) == [G1_GEN] and O.zkto_g1_is_on_curve(ptr(g)) == 1
    k = K["g1_double"]
    assert g1_from_arr(g1_add(g, g)) == [(int(k["x"]), int(k["y"]))]


def test_g1_multiples_and_add_table():   # g1_point.rs:315-345, 389-412
    g = g1_gen()
    pts = [(int(x), int(y)) for x, y in K["g1_multiples"]["points"]]
    for n in range(1, 11):
        assert g1_from_arr(g1_mul(g, n)) == [pts[n - 1]]
    for a, b, c in K["g1_add_table"]["cases"]:
        assert g1_from_arr(g1_add(g1_arr([pts[a - 1]]), g1_arr([pts[b - 1]]))) == [pts[c - 1]]


def test_g1_scalar_mul_kats():           # g1_point.rs:352-371 — scalar handed over as an Fq element
    g = g1_gen()
    for c in K["g1_scalar_mul"]["cases"]:
        assert g1_from_arr(g1_mul(g, int(c["k"]), limbs=6)) == [(int(c["x"]), int(c["y"]))]


def test_g1_infinity_cases():            # g1_point.rs:239-296, macros.rs:43-63
    g = g1_gen(); inf = g1_arr([None])
    neg = np.zeros_like(g); O.zkto_g1_neg_batch(ptr(g), ptr(neg), 1)
    assert g1_from_arr(g1_add(g, neg)) == [None]
    assert g1_from_arr(g1_add(g, inf)) == [G1_GEN] and g1_from_arr(g1_add(inf, g)) == [G1_GEN]
    assert g1_from_arr(g1_add(inf, inf)) == [None]
    assert g1_from_arr(g1_mul(g, 0)) == [None]                      # scalar 0 -> AtInfinity (Appendix C)
    assert g1_from_arr(g1_mul(g, R)) == [None]                      # order-r point, scalar used as-is
    neg_inf = np.zeros_like(g); O.zkto_g1_neg_batch(ptr(inf), ptr(neg_inf), 1)
    assert g1_from_arr(neg_inf) == [None]


def test_g1_vs_python_model():
    rng = SplitMix64(5)
    g = g1_gen()
    for _ in range(4):
        k = rng.below(R)
        assert g1_from_arr(g1_mul(g, k)) == [py_g1_mul(G1_GEN, k)]


def test_g1_msm_identity():              # polynomial.rs:1250-1285 shape: MSM == explicit sum
    rng = SplitMix64(6)
    ks = [rng.below(R) for _ in range(5)]
    ss = [rng.below(R) for _ in range(5)] ; ss[2] = 0
    bases = g1_arr([py_g1_mul(G1_GEN, k) for k in ks])
    sc = ints_to_arr(ss, 4); o = np.zeros((1, G1W), dtype=np.uint64)
    assert O.zkto_g1_msm(ptr(bases), ptr(sc), 4, 5, ptr(o)) == 0
    assert g1_from_arr(o) == [py_g1_mul(G1_GEN, sum(k * s for k, s in zip(ks, ss)) % R)]


# ---- G2 -----------------------------------------------------------------------
def g2_gen():
    g = np.zeros((1, G2W), dtype=np.uint64); O.zkto_g2_generator(ptr(g)); return g


def g2_mul(p, k, limbs=4):
    s = ints_to_arr([k], limbs); o = np.zeros((1, G2W), dtype=np.uint64)
    assert O.zkto_g2_mul_batch(ptr(p), ptr(s), limbs, ptr(o), 1, 1) == 0
    return o


def g2_add(p, q):
    o = np.zeros((1, G2W), dtype=np.uint64); assert O.zkto_g2_add_batch(ptr(p), ptr(q), ptr(o), 1) == 0; return o


def _g2pt(v):
    x1, x0, y1, y0 = (int(t) for t in v); return ((x1, x0), (y1, y0))


def test_g2_kats():                      # g2_point.rs:199-230, 320-350, 357-403, 421-444
    g = g2_gen()
    assert g2_from_arr(g) == [G2_GEN] and O.zkto_g2_is_on_curve(ptr(g)) == 1
    k = K["g2_double"]
    assert g2_from_arr(g2_add(g, g)) == [_g2pt([k["x_u1"], k["x_u0"], k["y_u1"], k["y_u0"]])]
    pts = [_g2pt(p) for p in K["g2_multiples"]["points"]]
    for n in range(1, 11):
        assert g2_from_arr(g2_mul(g, n)) == [pts[n - 1]]
    for a, b, c in K["g2_add_table"]["cases"]:
        assert g2_from_arr(g2_add(g2_arr([pts[a - 1]]), g2_arr([pts[b - 1]]))) == [pts[c - 1]]
    for c in K["g2_scalar_mul"]["cases"]:
        assert g2_from_arr(g2_mul(g, int(c["k"]))) == [_g2pt([c["x1"], c["x0"], c["y1"], c["y0"]])]
    neg = np.zeros_like(g); O.zkto_g2_neg_batch(ptr(g), ptr(neg), 1)
    assert g2_from_arr(g2_add(g, neg)) == [None]
    assert g2_from_arr(g2_mul(g, R)) == [None]


# ---- secp256k1 ------------------------------------------------------------------
def sp_arr(points):
    a = np.zeros((len(points), 9), dtype=np.uint64)
    for i, p in enumerate(points):
        if p is None: a[i, 8] = 1
        else: a[i, :4] = int_to_limbs(p[0], 4); a[i, 4:8] = int_to_limbs(p[1], 4)
    return a


def sp_from(a):
    return [None if int(r[8]) & 0xFFFFFFFF else (limbs_to_int(r[:4]), limbs_to_int(r[4:8])) for r in np.asarray(a).reshape(-1, 9)]


def test_secp_kats():                    # secp256k1/affine_point.rs:190-203, 292-311, 331-380, 383-424
    g = np.zeros((1, 9), dtype=np.uint64); O.zkto_secp_generator(ptr(g))
    def mul(p, k):
        s = ints_to_arr([k], 4); o = np.zeros((1, 9), dtype=np.uint64)
        assert O.zkto_secp_mul_batch(ptr(p), ptr(s), 4, ptr(o), 1, 1) == 0; return o
    def add(p, q):
        o = np.zeros((1, 9), dtype=np.uint64); assert O.zkto_secp_add_batch(ptr(p), ptr(q), ptr(o), 1) == 0; return o
    k = K["secp_double"]
    assert sp_from(add(g, g)) == [(int(k["x"]), int(k["y"]))]
    pts = [(int(x, 16), int(y, 16)) for x, y in K["secp_multiples"]["points"]]
    assert sp_from(g) == [pts[0]]
    for n in range(1, 11):
        assert sp_from(mul(g, n)) == [pts[n - 1]]
    for a, b, c in K["secp_add_table"]["cases"]:
        assert sp_from(add(sp_arr([pts[a - 1]]), sp_arr([pts[b - 1]]))) == [pts[c - 1]]
    for c in K["secp_scalar_mul"]["cases"]:
        assert sp_from(mul(g, int(c["k"], 16) % SECP_P)) == [(int(c["x"], 16), int(c["y"], 16))]
    k = K["secp_add_large"]
    p1, p2, p3 = [(int(k[n][0], 16), int(k[n][1], 16)) for n in ("p1", "p2", "p3")]
    assert sp_from(add(sp_arr([p1]), sp_arr([p2]))) == [p3]


# ---- row a3 through the four-field entry points (zkto_field_*) -------------------------------------------------
FIELDS = [(0, Q, 6), (1, R, 4), (2, SECP_P, 4), (3, SECP_N, 4)]


@pytest.mark.parametrize("field,mod,w", FIELDS)
def test_field_pow_cube_seq_vs_python_ints(field, mod, w):
    """pow / cube / pow_seq / repeat (prime_field_elem.rs:311-376) of the oracle against python integers, and the
    reference's own pow KATs (:888-909, :911-921) replayed in each field: all results are below every field order,
    so a^e mod p is the plain integer a^e and must agree with the reference's table after `% 10^8`."""
    rng = SplitMix64(77 + field)
    xs = [rng.below(mod) for _ in range(24)] + [0, 1, mod - 1, 2]
    es = [rng.below(1 << 256) for _ in range(24)] + [0, 0, (mod - 1) % (1 << 256), (mod - 2) % (1 << 256)]
    a, e = ints_to_arr(xs, w), ints_to_arr(es, 4)
    o = np.zeros_like(a)
    assert O.zkto_field_pow_batch(field, ptr(a), ptr(e), 4, 0, ptr(o), len(xs)) == 0
    assert arr_to_ints(o) == [pow(x, k, mod) for x, k in zip(xs, es)]
    assert O.zkto_field_pow_batch(field, ptr(a), ptr(e[5:6].copy()), 4, 1, ptr(o), len(xs)) == 0          # one shared exponent
    assert arr_to_ints(o) == [pow(x, es[5], mod) for x in xs]
    assert O.zkto_field_op(field, 6, ptr(a), None, ptr(o), len(xs), None) == 0                              # cube
    assert arr_to_ints(o) == [x * x % mod * x % mod for x in xs]
    seq = np.zeros((40, w), dtype=np.uint64)
    assert O.zkto_field_pow_seq(field, ptr(a[3:4].copy()), 40, ptr(seq), 0) == 0
    assert arr_to_ints(seq) == [pow(xs[3], i, mod) for i in range(40)]
    assert O.zkto_field_pow_seq(field, ptr(a[3:4].copy()), 7, ptr(seq), 1) == 0                             # repeat
    assert arr_to_ints(seq[:7]) == [xs[3]] * 7
    k = K["pow_various"]; m = int(k["order"])
    b = ints_to_arr([c[0] for c in k["cases"]], w); ex = ints_to_arr([c[1] for c in k["cases"]], 1); o = np.zeros_like(b)
    assert O.zkto_field_pow_batch(field, ptr(b), ptr(ex), 1, 0, ptr(o), len(b)) == 0
    assert [v % m for v in arr_to_ints(o)] == [c[2] for c in k["cases"]]
