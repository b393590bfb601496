"""GPU parity: every entry point of the C ABI (include/zkt.h) against the CPU oracle on the
same seeded inputs, bit-exact, plus the reference KATs straight through the GPU path and the
edge cases the reference tests (infinity, P+P, P+(-P), zero scalars, inverse of zero)."""
import ctypes, importlib
import numpy as np
import pytest
from zkt_testlib import *

pytestmark = pytest.mark.gpu
zk = importlib.import_module("zk-toolkit_amd")
O = oracle()
K = kats()


@pytest.fixture(scope="module")
def L():
    zk.init()
    return zk.lib()


def _rand_field(seed, n, mod, w):
    rng = SplitMix64(seed)
    xs = [rng.below(mod) for _ in range(n - 5)] + [0, 1, mod - 1, 2, mod - 2]
    return ints_to_arr(xs, w)


@pytest.mark.parametrize("pre,mod,w", [("fq", Q, 6), ("fr", R, 4)])
def test_field_batch_ops(L, pre, mod, w):
    n = 1000                                      # BASELINE config 1: 1000 Fq multiplications
    a, b = _rand_field(1, n, mod, w), _rand_field(101, n, mod, w)[::-1].copy()
    for op in ("add", "sub", "mul"):
        got, want = np.zeros_like(a), np.zeros_like(a)
        zk.check(getattr(L, f"zkt_{pre}_{op}_batch")(ptr(a), ptr(b), ptr(got), n))
        assert getattr(O, f"zkto_{pre}_{op}_batch")(ptr(a), ptr(b), ptr(want), n) == 0
        assert (got == want).all(), op
    for op in ("sqr", "neg"):
        got, want = np.zeros_like(a), np.zeros_like(a)
        zk.check(getattr(L, f"zkt_{pre}_{op}_batch")(ptr(a), ptr(got), n))
        assert getattr(O, f"zkto_{pre}_{op}_batch")(ptr(a), ptr(want), n) == 0
        assert (got == want).all(), op
    nz = a[(a != 0).any(axis=1)].copy()
    got, want = np.zeros_like(nz), np.zeros_like(nz)
    zk.check(getattr(L, f"zkt_{pre}_inv_batch")(ptr(nz), ptr(got), len(nz)))
    assert getattr(O, f"zkto_{pre}_inv_batch")(ptr(nz), ptr(want), len(nz), None) == 0
    assert (got == want).all()
    # inverse of zero: Err("Cannot find inverse of zero") (prime_field_elem.rs:379-382)
    rc = getattr(L, f"zkt_{pre}_inv_batch")(ptr(a), ptr(np.zeros_like(a)), n)
    assert rc == ZKT_ERR_INV_ZERO and L.zkt_last_error_index() == n - 5
    # empty batch
    zk.check(getattr(L, f"zkt_{pre}_mul_batch")(ptr(a), ptr(b), ptr(got), 0))


FIELD_IDS = {"fq": 0, "fr": 1, "sp": 2, "sn": 3}


@pytest.mark.parametrize("pre,mod,w", [("fq", Q, 6), ("fr", R, 4), ("sp", SECP_P, 4), ("sn", SECP_N, 4)])
def test_field_pow_cube_seq_repeat(L, pre, mod, w):
    """row a3: PrimeFieldElem::{pow, cube, pow_seq, repeat} (prime_field_elem.rs:311-376) for the four fields, against the oracle;
    the reference's pow KAT table (:888-909) straight through the GPU; a1/a2 for the two secp256k1 fields as well."""
    f = FIELD_IDS[pre]; n = 500
    a = _rand_field(7 + f, n, mod, w)
    rng = SplitMix64(900 + f)
    es = [rng.below(1 << (64 * k)) for k in (1, 2, 3, 4) for _ in range(n // 4 - 1)] + [0, 1, (mod - 1) % (1 << 256), (mod - 2) % (1 << 256)]
    e = ints_to_arr(es, 4)
    got, want = np.zeros_like(a), np.zeros_like(a)
    zk.check(getattr(L, f"zkt_{pre}_pow_batch")(ptr(a), ptr(e), 4, 0, ptr(got), n))
    assert O.zkto_field_pow_batch(f, ptr(a), ptr(e), 4, 0, ptr(want), n) == 0
    assert (got == want).all()
    zk.check(getattr(L, f"zkt_{pre}_pow_batch")(ptr(a), ptr(e[3:4].copy()), 4, 1, ptr(got), n))      # shared exponent
    assert O.zkto_field_pow_batch(f, ptr(a), ptr(e[3:4].copy()), 4, 1, ptr(want), n) == 0
    assert (got == want).all()
    zk.check(getattr(L, f"zkt_{pre}_cube_batch")(ptr(a), ptr(got), n))
    assert O.zkto_field_op(f, 6, ptr(a), None, ptr(want), n, None) == 0
    assert (got == want).all()
    m = 65536 + 3                                        # the range proof's y.pow_seq(n), bulletproofs.rs:87, at BASELINE config 5's length
    seq, wseq = np.zeros((m, w), dtype=np.uint64), np.zeros((m, w), dtype=np.uint64)
    zk.check(getattr(L, f"zkt_{pre}_pow_seq")(ptr(a[1:2].copy()), m, ptr(seq)))
    assert O.zkto_field_pow_seq(f, ptr(a[1:2].copy()), m, ptr(wseq), 0) == 0
    assert (seq == wseq).all()
    zk.check(getattr(L, f"zkt_{pre}_repeat")(ptr(a[1:2].copy()), 100, ptr(seq)))
    assert (seq[:100] == a[1]).all()
    k = K["pow_various"]; mm = int(k["order"])           # results below every field order: a^e mod p is the integer a^e
    b = ints_to_arr([c[0] for c in k["cases"]], w); ex = ints_to_arr([c[1] for c in k["cases"]], 1); o = np.zeros_like(b)
    zk.check(getattr(L, f"zkt_{pre}_pow_batch")(ptr(b), ptr(ex), 1, 0, ptr(o), len(b)))
    assert [v % mm for v in arr_to_ints(o)] == [c[2] for c in k["cases"]]
    zk.check(getattr(L, f"zkt_{pre}_pow_batch")(ptr(a), ptr(e), 4, 0, ptr(got), 0))                  # empty batch
    # a1/a2 on the same inputs (the secp256k1 fields have no other field-level test)
    b2 = _rand_field(55 + f, n, mod, w)[::-1].copy()
    for op, code in (("add", 0), ("sub", 1), ("mul", 2)):
        zk.check(getattr(L, f"zkt_{pre}_{op}_batch")(ptr(a), ptr(b2), ptr(got), n))
        assert O.zkto_field_op(f, code, ptr(a), ptr(b2), ptr(want), n, None) == 0
        assert (got == want).all(), op
    for op, code in (("sqr", 3), ("neg", 4)):
        zk.check(getattr(L, f"zkt_{pre}_{op}_batch")(ptr(a), ptr(got), n))
        assert O.zkto_field_op(f, code, ptr(a), None, ptr(want), n, None) == 0
        assert (got == want).all(), op
    nz = a[(a != 0).any(axis=1)].copy(); got, want = np.zeros_like(nz), np.zeros_like(nz)
    zk.check(getattr(L, f"zkt_{pre}_inv_batch")(ptr(nz), ptr(got), len(nz)))
    assert O.zkto_field_op(f, 5, ptr(nz), None, ptr(want), len(nz), None) == 0
    assert (got == want).all()


@pytest.mark.parametrize("pre,mod,w", [("fq", Q, 6), ("fr", R, 4), ("sp", SECP_P, 4), ("sn", SECP_N, 4)])
def test_field_inputs_are_reduced_like_prime_field_elem_new(L, pre, mod, w):
    """PrimeFieldElem::new reduces e mod order (prime_field_elem.rs:263-272): limb vectors at or above the order give the result of
    their residue, for every operand position."""
    top = 1 << (64 * w)
    raw = [mod, mod + 5, top - 1, 2 * mod + 1 if 2 * mod + 1 < top else mod + 1, 7]
    red = [x % mod for x in raw]
    a, ar = ints_to_arr(raw, w), ints_to_arr(red, w)
    b, br = ints_to_arr(raw[::-1], w), ints_to_arr(red[::-1], w)
    for op in ("add", "sub", "mul"):
        g1, g2 = np.zeros_like(a), np.zeros_like(a)
        zk.check(getattr(L, f"zkt_{pre}_{op}_batch")(ptr(a), ptr(b), ptr(g1), len(raw)))
        zk.check(getattr(L, f"zkt_{pre}_{op}_batch")(ptr(ar), ptr(br), ptr(g2), len(raw)))
        f = {"add": lambda x, y: (x + y) % mod, "sub": lambda x, y: (x - y) % mod, "mul": lambda x, y: x * y % mod}[op]
        assert arr_to_ints(g1) == arr_to_ints(g2) == [f(x, y) for x, y in zip(red, red[::-1])], op
    for op, f in (("sqr", lambda x: x * x % mod), ("neg", lambda x: -x % mod), ("cube", lambda x: pow(x, 3, mod))):
        g1 = np.zeros_like(a)
        zk.check(getattr(L, f"zkt_{pre}_{op}_batch")(ptr(a), ptr(g1), len(raw)))
        assert arr_to_ints(g1) == [f(x) for x in red], op
    g1 = np.zeros_like(a)
    rc = getattr(L, f"zkt_{pre}_inv_batch")(ptr(a), ptr(g1), len(raw))      # element 0 is the order itself = 0
    assert rc == ZKT_ERR_INV_ZERO and L.zkt_last_error_index() == 0
    zk.check(getattr(L, f"zkt_{pre}_inv_batch")(ptr(a[1:].copy()), ptr(g1[1:]), len(raw) - 1))
    assert arr_to_ints(g1[1:]) == [pow(x, -1, mod) for x in red[1:]]


def test_fq_mul_large_kat(L):                   # prime_field_elem.rs:600-617 is mod secp-n; here the Fq2..12 KAT inputs
    a1, b1 = Q - 3, Q - 5
    a = ints_to_arr([a1], 6); b = ints_to_arr([b1], 6); o = np.zeros_like(a)
    zk.check(L.zkt_fq_mul_batch(ptr(a), ptr(b), ptr(o), 1))
    assert arr_to_ints(o) == [15]


a1, b1, c1, d1 = Q - 3, Q - 5, Q - 7, Q - 9        # fq_test_helper.rs:9-34
a2, b2, c2, d2 = (a1, b1), (b1, c1), (c1, d1), (d1, a1)
a6, b6, c6, d6 = a2 + b2 + c2, b2 + c2 + d2, c2 + d2 + a2, d2 + a2 + b2


def _tw(L, name, w, x, y=None):
    a = ints_to_arr(list(x), 6).reshape(1, w); o = np.zeros((1, w), dtype=np.uint64)
    if y is not None:
        b = ints_to_arr(list(y), 6).reshape(1, w)
        zk.check(getattr(L, name)(ptr(a), ptr(b), ptr(o), 1))
    else:
        zk.check(getattr(L, name)(ptr(a), ptr(o), 1))
    return [str(v) for v in arr_to_ints(o.reshape(-1, 6))]


def test_tower_reference_kats_on_gpu(L):         # fq2.rs:166-226, fq6.rs:190-275, fq12.rs:198-329
    k = K["fq2"]; x, y = (a1, b1), (c1, d1)
    assert _tw(L, "zkt_fq2_add_batch", 12, x, y) == k["add"]
    assert _tw(L, "zkt_fq2_sub_batch", 12, x, y) == k["sub"]
    assert _tw(L, "zkt_fq2_mul_batch", 12, x, y) == k["mul"]
    assert _tw(L, "zkt_fq2_inv_batch", 12, x) == k["inv_a"]
    assert _tw(L, "zkt_fq2_inv_batch", 12, y) == k["inv_b"]
    assert _tw(L, "zkt_fq2_reduce_batch", 12, [int(v) for v in k["mul"]]) == k["reduce_mul"]
    k = K["fq6"]
    assert _tw(L, "zkt_fq6_add_batch", 36, a6, b6) == k["add"]
    assert _tw(L, "zkt_fq6_sub_batch", 36, a6, b6) == k["sub"]
    assert _tw(L, "zkt_fq6_mul_batch", 36, a6, b6) == k["mul"]
    assert _tw(L, "zkt_fq6_inv_batch", 36, a6) == k["inv_a"]
    assert _tw(L, "zkt_fq6_inv_batch", 36, b6) == k["inv_b"]
    assert _tw(L, "zkt_fq6_reduce_batch", 36, [int(v) for v in k["mul"]]) == k["reduce_mul"]
    k = K["fq12"]; a12, b12 = a6 + b6, c6 + d6
    assert _tw(L, "zkt_fq12_add_batch", 72, a12, b12) == k["add"]
    assert _tw(L, "zkt_fq12_sub_batch", 72, a12, b12) == k["sub"]
    assert _tw(L, "zkt_fq12_mul_batch", 72, a12, b12) == k["mul"]
    assert _tw(L, "zkt_fq12_inv_batch", 72, a12) == k["inv_a"]
    assert _tw(L, "zkt_fq12_inv_batch", 72, b12) == k["inv_b"]
    three = ints_to_arr([0] * 11 + [3], 6).reshape(1, 72); o = np.zeros((1, 72), dtype=np.uint64)
    e = np.array([4], dtype=np.uint32)
    zk.check(L.zkt_fq12_pow_batch(ptr(three), e.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 1, ptr(o), 1))
    assert arr_to_ints(o.reshape(-1, 6)) == [0] * 11 + [81]      # fq12.rs:198-206


@pytest.mark.parametrize("deg,w,fn", [(2, 12, "zkto_fq2_op"), (6, 36, "zkto_fq6_op"), (12, 72, "zkto_fq12_op")])
def test_tower_batch_vs_oracle(L, deg, w, fn):
    n = 96
    rng = SplitMix64(200 + deg)
    a = ints_to_arr([rng.below(Q) for _ in range(n * w // 6)], 6).reshape(n, w)
    b = ints_to_arr([rng.below(Q) for _ in range(n * w // 6)], 6).reshape(n, w)
    for op, name in ((0, "add"), (1, "sub"), (2, "mul")):
        got, want = np.zeros_like(a), np.zeros_like(a)
        zk.check(getattr(L, f"zkt_fq{deg}_{name}_batch")(ptr(a), ptr(b), ptr(got), n))
        assert getattr(O, fn)(op, ptr(a), ptr(b), ptr(want), n) == 0
        assert (got == want).all(), (deg, name)
    for op, name in ((3, "inv"), (4, "neg")):
        got, want = np.zeros_like(a), np.zeros_like(a)
        zk.check(getattr(L, f"zkt_fq{deg}_{name}_batch")(ptr(a), ptr(got), n))
        assert getattr(O, fn)(op, ptr(a), None, ptr(want), n) == 0
        assert (got == want).all(), (deg, name)
    z = np.zeros((2, w), dtype=np.uint64); z[0] = a[0]
    assert getattr(L, f"zkt_fq{deg}_inv_batch")(ptr(z), ptr(np.zeros_like(z)), 2) == ZKT_ERR_INV_ZERO
    assert L.zkt_last_error_index() == 1


def _gen(grp):
    g = np.zeros((1, {0: G1W, 1: G2W, 2: 9}[grp]), dtype=np.uint64)
    [O.zkto_g1_generator, O.zkto_g2_generator, O.zkto_secp_generator][grp](ptr(g))
    return g


GROUPS = [(0, "g1", G1W, R), (1, "g2", G2W, R), (2, "secp", 9, SECP_N)]


@pytest.mark.parametrize("grp,name,W,order", GROUPS)
def test_group_mul_and_add_vs_oracle(L, grp, name, W, order):
    n = 70 if grp != 1 else 40
    rng = SplitMix64(300 + grp)
    g = np.repeat(_gen(grp), n, axis=0)
    ks = [rng.below(order) for _ in range(n - 8)] + [0, 1, 2, order, order - 1, order + 1, (1 << 256) - 1, 3]
    sc = ints_to_arr(ks, 4)
    got, want = np.zeros_like(g), np.zeros_like(g)
    zk.check(getattr(L, f"zkt_{name}_mul_batch")(ptr(g), ptr(sc), 4, ptr(got), n))
    assert getattr(O, f"zkto_{name}_mul_batch")(ptr(g), ptr(sc), 4, ptr(want), n, 8) == 0
    assert (got == want).all()
    pts = want                                        # k_i * G, includes infinity rows (k = 0, order)
    # second operand: a rotation, the same points (P+P), negations (P+(-P)), infinities
    neg = np.zeros_like(pts)
    if grp < 2:
        zk.check(getattr(L, f"zkt_{name}_neg_batch")(ptr(pts), ptr(neg), n))
        wneg = np.zeros_like(pts); assert getattr(O, f"zkto_{name}_neg_batch")(ptr(pts), ptr(wneg), n) == 0
        assert (neg == wneg).all()
    else:
        sc2 = ints_to_arr([(order - k) % order for k in ks], 4)
        assert O.zkto_secp_mul_batch(ptr(g), ptr(sc2), 4, ptr(neg), n, 8) == 0
    for other in (np.roll(pts, 1, axis=0), pts, neg):
        got, want = np.zeros_like(pts), np.zeros_like(pts)
        zk.check(getattr(L, f"zkt_{name}_add_batch")(ptr(pts), ptr(other.copy()), ptr(got), n))
        assert getattr(O, f"zkto_{name}_add_batch")(ptr(pts), ptr(other.copy()), ptr(want), n) == 0
        assert (got == want).all()


def test_g1_reference_kats_on_gpu(L):            # g1_point.rs:224-237, 315-345, 352-371, 389-412
    g = _gen(0)
    pts = [(int(x), int(y)) for x, y in K["g1_multiples"]["points"]]
    gs = np.repeat(g, 10, axis=0); sc = ints_to_arr(list(range(1, 11)), 4); got = np.zeros_like(gs)
    zk.check(L.zkt_g1_mul_batch(ptr(gs), ptr(sc), 4, ptr(got), 10))
    assert g1_from_arr(got) == pts
    cases = K["g1_scalar_mul"]["cases"]                # scalar handed over as an Fq element (6 limbs)
    gs = np.repeat(g, len(cases), axis=0); sc = ints_to_arr([int(c["k"]) for c in cases], 6); got = np.zeros_like(gs)
    zk.check(L.zkt_g1_mul_batch(ptr(gs), ptr(sc), 6, ptr(got), len(cases)))
    assert g1_from_arr(got) == [(int(c["x"]), int(c["y"])) for c in cases]
    tab = K["g1_add_table"]["cases"]
    a = g1_arr([pts[t[0] - 1] for t in tab]); b = g1_arr([pts[t[1] - 1] for t in tab]); got = np.zeros_like(a)
    zk.check(L.zkt_g1_add_batch(ptr(a), ptr(b), ptr(got), len(tab)))
    assert g1_from_arr(got) == [pts[t[2] - 1] for t in tab]


def test_g2_and_secp_reference_kats_on_gpu(L):   # g2_point.rs:320-350,357-403; secp256k1/affine_point.rs:292-311,331-380
    g = _gen(1)
    pts = [((int(p[0]), int(p[1])), (int(p[2]), int(p[3]))) for p in K["g2_multiples"]["points"]]
    gs = np.repeat(g, 11, axis=0); c = K["g2_scalar_mul"]["cases"][0]
    sc = ints_to_arr(list(range(1, 11)) + [int(c["k"])], 4); got = np.zeros_like(gs)
    zk.check(L.zkt_g2_mul_batch(ptr(gs), ptr(sc), 4, ptr(got), 11))
    assert g2_from_arr(got) == pts + [((int(c["x1"]), int(c["x0"])), (int(c["y1"]), int(c["y0"])))]
    g = _gen(2)
    cases = K["secp_scalar_mul"]["cases"]
    gs = np.repeat(g, len(cases), axis=0); sc = ints_to_arr([int(c["k"], 16) % SECP_P for c in cases], 4); got = np.zeros_like(gs)
    zk.check(L.zkt_secp_mul_batch(ptr(gs), ptr(sc), 4, ptr(got), len(cases)))
    for row, c in zip(got, cases):
        assert (limbs_to_int(row[:4]), limbs_to_int(row[4:8]), int(row[8])) == (int(c["x"], 16), int(c["y"], 16), 0)


def test_tate_batch_vs_oracle_and_bilinearity(L):
    # BASELINE config 1: 10 Tate pairings, P_i = k_i G1, Q_i = k'_i G2 (g1_point.rs:83-88)
    n = 10
    rng = SplitMix64(2)
    ks = [rng.below(R) for _ in range(n)]; ks[0] = 1
    kq = [rng.below(R) for _ in range(n)]; kq[0] = 1
    ps, qs = np.repeat(_gen(0), n, axis=0), np.repeat(_gen(1), n, axis=0)
    P, Qp = np.zeros_like(ps), np.zeros_like(qs)
    assert O.zkto_g1_mul_batch(ptr(ps), ptr(ints_to_arr(ks, 4)), 4, ptr(P), n, 8) == 0
    assert O.zkto_g2_mul_batch(ptr(qs), ptr(ints_to_arr(kq, 4)), 4, ptr(Qp), n, 8) == 0
    got, want = np.zeros((n, FQ12), dtype=np.uint64), np.zeros((n, FQ12), dtype=np.uint64)
    zk.check(L.zkt_tate_batch(ptr(P), ptr(Qp), ptr(got), n))
    assert O.zkto_pairing_batch(3, ptr(P), ptr(Qp), ptr(want), n, 8, None) == 0
    assert (got == want).all()
    # bilinearity on the GPU results alone (pairing.rs:107-123): e(P,Q)^(k k') == e(kP, k'Q) via GT mul chain is
    # too long; check e(P+10P,Q) = e(P,Q) e(10P,Q) as the reference does
    g1 = _gen(0); p10 = np.zeros_like(g1); p11 = np.zeros_like(g1)
    assert O.zkto_g1_mul_batch(ptr(g1), ptr(ints_to_arr([10], 4)), 4, ptr(p10), 1, 1) == 0
    assert O.zkto_g1_add_batch(ptr(g1), ptr(p10), ptr(p11), 1) == 0
    three = np.concatenate([g1, p10, p11]); q3 = np.repeat(_gen(1), 3, axis=0); e = np.zeros((3, FQ12), dtype=np.uint64)
    zk.check(L.zkt_tate_batch(ptr(three), ptr(q3), ptr(e), 3))
    prod = np.zeros((1, FQ12), dtype=np.uint64)
    zk.check(L.zkt_fq12_mul_batch(ptr(e[0:1].copy()), ptr(e[1:2].copy()), ptr(prod), 1))
    assert L.zkt_gt_eq(ptr(prod), ptr(e[2:3].copy())) == 1
    # infinity argument -> error with the index (rational_function.rs:36,59 panic)
    P[3, :] = 0; P[3, 12] = 1
    assert L.zkt_tate_batch(ptr(P), ptr(Qp), ptr(got), n) == ZKT_ERR_INFINITY and L.zkt_last_error_index() == 3


def _msm_oracle(bases, sc, n):
    o = np.zeros((1, G1W), dtype=np.uint64)
    assert O.zkto_g1_msm(ptr(bases), ptr(sc), 4, n, ptr(o)) == 0
    return o


@pytest.mark.parametrize("n", [1, 2, 5, 33, 255, 1024])
def test_g1_msm_vs_oracle(L, n):
    rng = SplitMix64(400 + n)
    g = np.repeat(_gen(0), n, axis=0); bases = np.zeros_like(g)
    zk.check(L.zkt_g1_mul_batch(ptr(g), ptr(ints_to_arr([rng.below(R) for _ in range(n)], 4)), 4, ptr(bases), n))
    ss = [rng.below(R) for _ in range(n)]
    if n > 4: ss[1] = 0; ss[2] = 1; ss[3] = R - 1
    sc = ints_to_arr(ss, 4)
    got = np.zeros((1, G1W), dtype=np.uint64)
    zk.check(L.zkt_g1_msm(ptr(bases), ptr(sc), n, ptr(got)))
    assert (got == _msm_oracle(bases, sc, n)).all()


def test_g1_msm_edge_cases(L):
    # repeated bases (bulletproofs.rs:231-246 uses gg=[g,g]), a base and its negation, infinity bases,
    # all-zero scalars, scalars >= r used as-is (macros.rs:10-21)
    g = _gen(0); n = 8
    neg = np.zeros_like(g); assert O.zkto_g1_neg_batch(ptr(g), ptr(neg), 1) == 0
    inf = g1_arr([None])
    bases = np.concatenate([g, g, g, neg, inf, g, neg, inf])
    for ss in ([1, 1, 1, 1, 5, 0, 0, 7], [5, 5, 3, 13, 1, 2, 2, 0], [0] * 8, [R, R + 1, (1 << 256) - 1, 1, 1, 2**255, 3, 4]):
        sc = ints_to_arr(ss, 4); got = np.zeros((1, G1W), dtype=np.uint64)
        zk.check(L.zkt_g1_msm(ptr(bases), ptr(sc), n, ptr(got)))
        assert (got == _msm_oracle(bases, sc, n)).all(), ss
    got = np.zeros((1, G1W), dtype=np.uint64)
    zk.check(L.zkt_g1_msm(ptr(bases), ptr(sc), 0, ptr(got)))
    assert g1_from_arr(got) == [None]                 # empty sum = G1Point::zero() (polynomial.rs:275)


def test_raw_miller_values_and_weil_vs_oracle(L):    # pairing.rs:54-55,75-84; bilinearity of weil as in pairing.rs:107-151
    n = 4
    rng = SplitMix64(700)
    P, Qp = np.zeros((n, G1W), np.uint64), np.zeros((n, G2W), np.uint64)
    assert O.zkto_g1_mul_batch(ptr(np.repeat(_gen(0), n, axis=0)), ptr(ints_to_arr([1] + [rng.below(R) for _ in range(n - 1)], 4)), 4, ptr(P), n, 4) == 0
    assert O.zkto_g2_mul_batch(ptr(np.repeat(_gen(1), n, axis=0)), ptr(ints_to_arr([1] + [rng.below(R) for _ in range(n - 1)], 4)), 4, ptr(Qp), n, 4) == 0
    for which, fn, swap in ((0, L.zkt_miller_g1g2_batch, False), (1, L.zkt_miller_g2g1_batch, True), (2, L.zkt_weil_batch, False)):
        got, want = np.zeros((n, FQ12), np.uint64), np.zeros((n, FQ12), np.uint64)
        zk.check(fn(ptr(Qp), ptr(P), ptr(got), n) if swap else fn(ptr(P), ptr(Qp), ptr(got), n))
        assert O.zkto_pairing_batch(which, ptr(P), ptr(Qp), ptr(want), n, 8, None) == 0
        assert (got == want).all(), which
    inf = P.copy(); inf[2, :] = 0; inf[2, 12] = 1
    assert L.zkt_weil_batch(ptr(inf), ptr(Qp), ptr(got), n) == ZKT_ERR_INFINITY and L.zkt_last_error_index() == 2


@pytest.mark.parametrize("kind", ["ones", "small", "equal", "few_big"])
def test_g1_msm_skewed_scalars(L, kind):
    """Skewed scalar distributions put thousands of terms in one bucket (a real witness is full of 0/1): the hot
    buckets are split into <=128-entry tasks and merged; the sum must not change."""
    n = 4096
    rng = SplitMix64(900)
    g = np.repeat(_gen(0), n, axis=0); bases = np.zeros_like(g)
    zk.check(L.zkt_g1_mul_batch(ptr(g), ptr(ints_to_arr([rng.below(R) for _ in range(n)], 4)), 4, ptr(bases), n))
    if kind == "ones": ss = [1] * n
    elif kind == "small": ss = [rng.below(4) for _ in range(n)]
    elif kind == "equal": ss = [rng.below(R)] * n
    else: ss = [1] * n; ss[7] = R - 1; ss[100] = (1 << 256) - 1; ss[4095] = 0
    sc = ints_to_arr(ss, 4)
    got = np.zeros((1, G1W), dtype=np.uint64)
    zk.check(L.zkt_g1_msm(ptr(bases), ptr(sc), n, ptr(got)))
    if kind == "equal":                                # sum_i s*P_i = s * sum_i P_i : keeps the oracle side cheap
        acc = np.zeros((1, G1W), np.uint64); assert O.zkto_g1_msm(ptr(bases), ptr(ints_to_arr([1] * n, 4)), 4, n, ptr(acc)) == 0
        want = np.zeros((1, G1W), np.uint64); assert O.zkto_g1_mul_batch(ptr(acc), ptr(sc[:1].copy()), 4, ptr(want), 1, 1) == 0
    else:
        want = _msm_oracle(bases, sc, n)
    assert (got == want).all(), kind


def test_plain_c_consumer_of_the_abi(L, tmp_path):
    """tests/c/abi_consumer.c: a C99 program linked against libzkt_hip.so only — no Python or torch in the calling process."""
    import subprocess
    exe = str(tmp_path / "abi_consumer")
    libdir = os.path.join(ROOT, "zk-toolkit_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_consumer.c"),
                           "-L", libdir, "-lzkt_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "abi_consumer ok" in out.stdout, out.stdout + out.stderr


def test_fq_lazy_limb_programs_on_device(L):
    """The self-test program of the lazily reduced Fq arithmetic (csrc/fq_program.h) compiled for the DEVICE: 256 lanes x 3000 steps of
    add/sub/mul/fused ops over non-canonical representatives must keep the limb/size invariants and match plain mod-p arithmetic
    (the host build of the same header is checked by tests/test_hostcheck.py; the model is shared with it)."""
    from test_hostcheck import _fq_program_model
    count, steps, seed0 = 256, 3000, 424242
    rng = SplitMix64(8080)
    edge = [0, 1, Q - 1, Q - 2, 2, (Q + 1) // 2, (1 << 380), (1 << 364) - 1, Q - (1 << 364), (1 << 28) - 1, Q >> 1]
    regs = [[edge[(i + k) % len(edge)] if (i + k) % 5 == 0 else rng.below(Q) for k in range(4)] for i in range(count)]
    a = ints_to_arr([v for r in regs for v in r], 6); o = np.zeros_like(a); bad = np.zeros(count, np.int32)
    L.zkt_selftest_fq_program.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    zk.check(L.zkt_selftest_fq_program(seed0, steps, a.ctypes.data, o.ctypes.data, bad.ctypes.data, count))
    assert not bad.any(), f"invariant violations on lanes {np.nonzero(bad)[0][:8]}"
    got = arr_to_ints(o)
    for i in range(0, count, 1):
        assert got[4 * i:4 * i + 4] == _fq_program_model(seed0 + i, steps, regs[i]), i


def test_pairing_g1_argument_outside_the_subgroup(L):
    """Degenerate pairing inputs (round-1 review, weak #8): P on the curve but outside G1.  The reference either panics — a multiple of P met
    by its binary chain is infinity (rational_function.rs:36) — or returns a value that depends on P's order; the engine must do the same,
    element by element, inside an otherwise honest batch."""
    rng = SplitMix64(4711)
    g1 = np.zeros((1, G1W), np.uint64); O.zkto_g1_generator(ptr(g1))
    g2 = np.zeros((1, G2W), np.uint64); O.zkto_g2_generator(ptr(g2))
    n = 70
    P = np.zeros((n, G1W), np.uint64); Qs = np.zeros((n, G2W), np.uint64)
    zk.check(L.zkt_g1_mul_batch(ptr(np.repeat(g1, n, axis=0)), ptr(ints_to_arr([rng.below(R - 1) + 1 for _ in range(n)], 4)), 4, ptr(P), n))
    zk.check(L.zkt_g2_mul_batch(ptr(np.repeat(g2, n, axis=0)), ptr(ints_to_arr([rng.below(R - 1) + 1 for _ in range(n)], 4)), 4, ptr(Qs), n))
    deg = degenerate_g1_points()
    value_cases = [pt for _, pt in deg if O.zkto_pairing_batch(3, ptr(g1_arr([pt])), ptr(Qs[:1].copy()), ptr(np.zeros((1, FQ12), np.uint64)), 1, 1, None) == 0]
    panic_cases = [pt for _, pt in deg if pt not in value_cases]
    assert value_cases and panic_cases
    # (1) points outside G1 on which the reference returns a value: the whole batch matches the oracle
    for k, pt in enumerate(value_cases): P[3 + 29 * k] = g1_arr([pt])[0]
    got, want = np.zeros((n, FQ12), np.uint64), np.zeros((n, FQ12), np.uint64)
    zk.check(L.zkt_tate_batch(ptr(P), ptr(Qs), ptr(got), n))
    assert O.zkto_pairing_batch(3, ptr(P), ptr(Qs), ptr(want), n, 16, None) == 0
    assert (got == want).all()
    graw, wraw = np.zeros((n, FQ12), np.uint64), np.zeros((n, FQ12), np.uint64)
    zk.check(L.zkt_miller_g1g2_batch(ptr(P), ptr(Qs), ptr(graw), n))
    assert O.zkto_pairing_batch(0, ptr(P), ptr(Qs), ptr(wraw), n, 16, None) == 0
    assert (graw == wraw).all()
    # on-curve / subgroup predicates tell these points apart
    oc, sg = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    zk.check(L.zkt_g1_is_on_curve_batch(ptr(P), oc.ctypes.data, n)); zk.check(L.zkt_g1_in_subgroup_batch(ptr(P), sg.ctypes.data, n))
    assert oc.all() and sg.sum() == n - len(value_cases) and all(sg[3 + 29 * k] == 0 for k in range(len(value_cases)))
    # The deciding entry points evaluate such elements the reference's way — lhs = tate(P0, Q0), rhs = tate(P1, Q1), Fq12 equality (signature.rs:34-39) — where the
    # product form "tate(P0,Q0) tate(-P1,Q1) == 1" is only valid for points of order r.  Element i compares (P_i, Q_i) with (P_i, Q_i) when i is even (equal sides: accept,
    # for ANY point the reference gives a value on) and with (P_{i+1}, Q_{i+1}) when i is odd (different pairs: reject), so the degenerate points (odd positions 3 + 29 k,
    # and their even neighbours' partners) meet both outcomes; the oracle's GT values decide what is expected.
    g1s, g2s = np.zeros((2 * n, G1W), np.uint64), np.zeros((2 * n, G2W), np.uint64)
    for i in range(n):
        j = i if i % 2 == 0 else (i + 1) % n
        g1s[2 * i], g2s[2 * i], g1s[2 * i + 1], g2s[2 * i + 1] = P[i], Qs[i], P[j], Qs[j]
    expect = [int((want[i] == want[i if i % 2 == 0 else (i + 1) % n]).all()) for i in range(n)]
    # a degenerate point on the RIGHT side too: element 4 compares (P_4, Q_4) with (P_3, Q_3) — P_3 is outside G1
    g1s[2 * 4 + 1], g2s[2 * 4 + 1] = P[3], Qs[3]; expect[4] = int((want[4] == want[3]).all())
    ok = np.zeros(n, np.uint32)
    zk.check(L.zkt_pairing_product_check_batch(ptr(g1s), ptr(g2s), (ctypes.c_uint8 * 2)(0, 1), 2, n, ok.ctypes.data))
    assert [int(v) for v in ok] == expect and sum(expect) == n // 2 - 1      # every even element but the fourth
    # ... and the same elements with the left pair repeated on the right: equal sides, accepted whatever the order of P
    zk.check(L.zkt_pairing_product_check_batch(ptr(np.repeat(P, 2, axis=0)), ptr(np.repeat(Qs, 2, axis=0)), (ctypes.c_uint8 * 2)(0, 1), 2, n, ok.ctypes.data))
    assert ok.all()
    # fail-closed mode (the behaviour of earlier versions, kept behind an explicit switch): an element with a G1 argument outside the subgroup is rejected outright
    L.zkt_verify_set_fail_closed(1)
    try:
        zk.check(L.zkt_pairing_product_check_batch(ptr(np.repeat(P, 2, axis=0)), ptr(np.repeat(Qs, 2, axis=0)), (ctypes.c_uint8 * 2)(0, 1), 2, n, ok.ctypes.data))
        assert [int(v) for v in ok] == [int(v) for v in sg]
    finally:
        L.zkt_verify_set_fail_closed(0)
    # (2) points on which the reference panics: ZKT_ERR_INFINITY with the index of the first such element
    for k, pt in enumerate(panic_cases): P[40 + 7 * k] = g1_arr([pt])[0]
    assert L.zkt_tate_batch(ptr(P), ptr(Qs), ptr(got), n) == ZKT_ERR_INFINITY and L.zkt_last_error_index() == 40
    idx = ctypes.c_size_t(0)
    assert O.zkto_pairing_batch(3, ptr(P), ptr(Qs), ptr(want), n, 1, ctypes.byref(idx)) == ZKT_ERR_INFINITY and idx.value == 40
    assert L.zkt_miller_g1g2_batch(ptr(P), ptr(Qs), ptr(graw), n) == ZKT_ERR_INFINITY and L.zkt_last_error_index() == 40
    # the deciding entry points report the reference's panic the same way (first such element)
    assert L.zkt_pairing_product_check_batch(ptr(np.repeat(P, 2, axis=0)), ptr(np.repeat(Qs, 2, axis=0)), (ctypes.c_uint8 * 2)(0, 1), 2, n, ok.ctypes.data) == ZKT_ERR_INFINITY
    assert L.zkt_last_error_index() == 40


def test_generators_and_curve_predicates(L):
    """a16: generators (g1_point.rs:38-59, g2_point.rs:36-58, secp256k1/affine_point.rs:40-60) and is_rational_point (g1_point.rs:97-113)"""
    for name, W, gen in (("g1", G1W, O.zkto_g1_generator), ("g2", G2W, O.zkto_g2_generator), ("secp", 9, O.zkto_secp_generator)):
        got, want = np.zeros((1, W), np.uint64), np.zeros((1, W), np.uint64)
        getattr(L, f"zkt_{name}_generator")(ptr(got)); gen(ptr(want))
        assert (got == want).all(), name
        pts = np.repeat(got, 6, axis=0)
        ks = ints_to_arr([1, 2, 12345, R - 1, SECP_N - 1, 7], 4)
        zk.check(getattr(L, f"zkt_{name}_mul_batch")(ptr(np.repeat(got, 6, axis=0)), ptr(ks), 4, ptr(pts), 6))
        pts[4, 0] ^= np.uint64(1)                                   # x tampered: off the curve
        pts[5] = 0; pts[5, W - 1] = 1                               # infinity: not a rational point; trivially in the subgroup
        oc, sg = np.zeros(6, np.uint32), np.zeros(6, np.uint32)
        zk.check(getattr(L, f"zkt_{name}_is_on_curve_batch")(ptr(pts), oc.ctypes.data, 6))
        zk.check(getattr(L, f"zkt_{name}_in_subgroup_batch")(ptr(pts), sg.ctypes.data, 6))
        assert oc.tolist() == [1, 1, 1, 1, 0, 0], name
        assert sg.tolist()[:4] == [1, 1, 1, 1] and sg[5] == 1, name


def test_calls_from_several_threads(L):
    """include/zkt.h, Threading: entry points may be called from any thread (each selects the library's device); independent batch calls are
    serialised on the staging stream, calls on one handle by its lock.  Four threads hammer different entry points and two of them share ONE
    resident base set; every result must equal the single-threaded one."""
    import threading, torch
    n = 2000
    rng = np.random.Generator(np.random.PCG64(9))
    ks = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64); ks[:, 3] >>= np.uint64(2)
    ss = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64); ss[:, 3] >>= np.uint64(2)
    g = np.zeros((1, G1W), np.uint64); O.zkto_g1_generator(ptr(g))
    bases = np.zeros((n, G1W), np.uint64)
    zk.check(L.zkt_g1_mul_batch(ptr(np.repeat(g, n, axis=0)), ptr(ks), 4, ptr(bases), n))
    h = ctypes.c_void_p(); zk.check(L.zkt_g1_bases_upload(ptr(bases), n, ctypes.byref(h)))
    d_s = torch.from_numpy(ss.view(np.int64)).cuda()
    want_msm = np.zeros((1, G1W), np.uint64)
    zk.check(L.zkt_g1_msm_dev(h, ctypes.c_void_p(d_s.data_ptr()), n, None, ptr(want_msm), None))
    a, b = _rand_field(3, 800, Q, 6), _rand_field(4, 800, Q, 6)
    want_mul = np.zeros_like(a); zk.check(L.zkt_fq_mul_batch(ptr(a), ptr(b), ptr(want_mul), 800))
    P = bases[:3].copy(); Qs = np.zeros((3, G2W), np.uint64)
    g2 = np.zeros((1, G2W), np.uint64); O.zkto_g2_generator(ptr(g2))
    zk.check(L.zkt_g2_mul_batch(ptr(np.repeat(g2, 3, axis=0)), ptr(ks[:3].copy()), 4, ptr(Qs), 3))
    want_e = np.zeros((3, FQ12), np.uint64); zk.check(L.zkt_tate_batch(ptr(P), ptr(Qs), ptr(want_e), 3))
    errors = []

    def msm_worker():
        try:
            for _ in range(6):
                got = np.zeros((1, G1W), np.uint64)
                zk.check(L.zkt_g1_msm_dev(h, ctypes.c_void_p(d_s.data_ptr()), n, None, ptr(got), None))
                assert (got == want_msm).all()
        except Exception as e: errors.append(("msm", repr(e)))

    def mul_worker():
        try:
            for _ in range(20):
                got = np.zeros_like(a); zk.check(L.zkt_fq_mul_batch(ptr(a), ptr(b), ptr(got), 800)); assert (got == want_mul).all()
        except Exception as e: errors.append(("mul", repr(e)))

    def tate_worker():
        try:
            for _ in range(3):
                got = np.zeros((3, FQ12), np.uint64); zk.check(L.zkt_tate_batch(ptr(P), ptr(Qs), ptr(got), 3)); assert (got == want_e).all()
        except Exception as e: errors.append(("tate", repr(e)))

    ts = [threading.Thread(target=f) for f in (msm_worker, msm_worker, mul_worker, tate_worker)]
    for t in ts: t.start()
    for t in ts: t.join(timeout=300)
    L.zkt_g1_bases_free(h)
    assert not errors, errors
    assert not any(t.is_alive() for t in ts)


def test_pairing_g2_argument_outside_the_subgroup(L):
    """The 127-step loop needs Q in G2 and both points on their curves; Q on the twist outside G2 falls back to the 255-step loop, a point off its
    curve to the reference's own chain — element by element inside an honest batch, values identical to the oracle's reference algorithm."""
    rng = SplitMix64(4712)
    g1 = np.zeros((1, G1W), np.uint64); O.zkto_g1_generator(ptr(g1))
    g2 = np.zeros((1, G2W), np.uint64); O.zkto_g2_generator(ptr(g2))
    n = 70
    P = np.zeros((n, G1W), np.uint64); Qs = np.zeros((n, G2W), np.uint64)
    zk.check(L.zkt_g1_mul_batch(ptr(np.repeat(g1, n, axis=0)), ptr(ints_to_arr([rng.below(R - 1) + 1 for _ in range(n)], 4)), 4, ptr(P), n))
    zk.check(L.zkt_g2_mul_batch(ptr(np.repeat(g2, n, axis=0)), ptr(ints_to_arr([rng.below(R - 1) + 1 for _ in range(n)], 4)), 4, ptr(Qs), n))
    outside = {5: to_abi_g2(py_twist_point(rng)), 37: to_abi_g2(py_twist_point(rng)), 64: to_abi_g2(py_twist_point(rng))}     # on E', outside G2
    for i, pt in outside.items(): Qs[i] = g2_arr([pt])[0]
    (x1, x0), (y1, y0) = g2_from_arr(Qs[11:12])[0]
    Qs[11] = g2_arr([((x1, x0), (y1, (y0 + 1) % Q))])[0]                                                                       # off the twist
    px, py_ = g1_from_arr(P[50:51])[0]
    p_off = g1_arr([(px, (py_ + 1) % Q)])
    off_p_has_value = O.zkto_pairing_batch(3, ptr(p_off), ptr(Qs[50:51].copy()), ptr(np.zeros((1, FQ12), np.uint64)), 1, 1, None) == 0
    if off_p_has_value: P[50] = p_off[0]                                                                                        # off the curve of G1
    got, want = np.zeros((n, FQ12), np.uint64), np.zeros((n, FQ12), np.uint64)
    zk.check(L.zkt_tate_batch(ptr(P), ptr(Qs), ptr(got), n))
    assert O.zkto_pairing_batch(3, ptr(P), ptr(Qs), ptr(want), n, 16, None) == 0
    assert (got == want).all()
    oc, sg = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    zk.check(L.zkt_g2_is_on_curve_batch(ptr(Qs), oc.ctypes.data, n)); zk.check(L.zkt_g2_in_subgroup_batch(ptr(Qs), sg.ctypes.data, n))
    assert [i for i in range(n) if not oc[i]] == [11] and [i for i in range(n) if not sg[i]] == [5, 11, 37, 64]
    # the fused product check takes the same routes: e(P,Q) e(-P,Q) == 1 holds for P in G1 and ANY Q on the twist (255-step kernel behind the 127-step one)
    ok = np.zeros(n, np.uint32)
    zk.check(L.zkt_pairing_product_check_batch(ptr(np.repeat(P, 2, axis=0)), ptr(np.repeat(Qs, 2, axis=0)), (ctypes.c_uint8 * 2)(0, 1), 2, n, ok.ctypes.data))
    # (element 50, P off its curve: both sides are the same pair, and the reference's own evaluation — which such an element now gets — finds them equal; round 3 failed it closed)
    assert all(int(ok[i]) == 1 for i in range(n) if i not in (11, 50)) and (not off_p_has_value or int(ok[50]) == 1)


@pytest.mark.parametrize("pre,mod,w", [("fq", Q, 6), ("fr", R, 4), ("sp", SECP_P, 4), ("sn", SECP_N, 4)])
def test_field_vector_sum_and_scalar_broadcast(L, pre, mod, w):
    """row a18's vector forms: PrimeFieldElems::sum (prime_field_elems.rs:35-41) and PrimeFieldElems * PrimeFieldElem (:152-175) against the oracle's
    element operations (the fold acc + x; x * k per element), for one-block, two-pass and ragged sizes; the reference's empty-vector asserts."""
    f = FIELD_IDS[pre]
    sz, vp = ctypes.c_size_t, ctypes.c_void_p
    getattr(L, f"zkt_{pre}_sum").argtypes = [vp, sz, vp]; getattr(L, f"zkt_{pre}_scale_batch").argtypes = [vp, vp, vp, sz]
    O.zkto_field_op.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, sz, vp]
    for n in (1, 2, 255, 257, 1000, 70001):
        a = _rand_field(900 + n, max(n, 6), mod, w)[:n].copy()
        if n > 3: a[3] = int_to_limbs((1 << (64 * w)) - 1, w)                      # not below the order: reduced on load like PrimeFieldElem::new
        got = np.zeros((1, w), np.uint64)
        zk.check(getattr(L, f"zkt_{pre}_sum")(a.ctypes.data, n, got.ctypes.data))
        if n <= 1000:                                                               # the oracle's fold, element by element
            acc = np.zeros((1, w), np.uint64)
            for i in range(n):
                nxt = np.zeros((1, w), np.uint64)
                assert O.zkto_field_op(f, 0, acc.ctypes.data, a[i:i + 1].copy().ctypes.data, nxt.ctypes.data, 1, None) == 0
                acc = nxt
            assert (got == acc).all(), (pre, n)
        assert limbs_to_int(got[0]) == sum(limbs_to_int(r) for r in a) % mod, (pre, n)
        k = ints_to_arr([SplitMix64(n).below(mod) if n != 2 else (1 << (64 * w)) - 1], w)
        got, want = np.zeros_like(a), np.zeros_like(a)
        zk.check(getattr(L, f"zkt_{pre}_scale_batch")(a.ctypes.data, k.ctypes.data, got.ctypes.data, n))
        assert O.zkto_field_op(f, 2, a.ctypes.data, np.repeat(k, n, axis=0).ctypes.data, want.ctypes.data, n, None) == 0
        assert (got == want).all(), (pre, n)
    one = np.zeros((1, w), np.uint64)
    assert getattr(L, f"zkt_{pre}_sum")(one.ctypes.data, 0, one.ctypes.data) == ZKT_ERR_SHAPE           # assert!(self.0.len() > 0)
    assert getattr(L, f"zkt_{pre}_scale_batch")(one.ctypes.data, one.ctypes.data, one.ctypes.data, 0) == ZKT_ERR_SHAPE


@pytest.mark.parametrize("grp,name,W,order", GROUPS)
def test_point_vector_sum_and_scalar_broadcast(L, grp, name, W, order):
    """AffinePoints::sum (secp256k1/affine_points.rs:25-31: the fold from zero) and AffinePoints * PrimeFieldElem (:105-122) against the oracle's affine
    additions / scalar multiplications — with infinities, repeated points (P + P) and opposite points (P + (-P)) inside the vector."""
    sz, vp = ctypes.c_size_t, ctypes.c_void_p
    getattr(L, f"zkt_{name}_sum").argtypes = [vp, sz, vp]; getattr(L, f"zkt_{name}_scale_batch").argtypes = [vp, vp, ctypes.c_int, vp, sz]
    rng = SplitMix64(7100 + grp)
    n = 150 if grp != 1 else 90
    ks = [rng.below(order) for _ in range(n - 6)]
    ks += [0, ks[0], (order - ks[1]) % order, 1, order - 1, 0]                       # infinity, a repeat of point 0, the opposite of point 1, G, -G, infinity
    g = np.repeat(_gen(grp), n, axis=0)
    pts = np.zeros_like(g)
    assert getattr(O, f"zkto_{name}_mul_batch")(ptr(g), ptr(ints_to_arr(ks, 4)), 4, ptr(pts), n, 8) == 0
    for cnt in (0, 1, 2, 64, 65, n):
        got = np.zeros((1, W), np.uint64)
        zk.check(getattr(L, f"zkt_{name}_sum")(pts.ctypes.data, cnt, got.ctypes.data))
        acc = np.zeros((1, W), np.uint64); acc[0, W - 1] = 1                         # AffinePoint::zero()
        for i in range(cnt):
            nxt = np.zeros((1, W), np.uint64)
            assert getattr(O, f"zkto_{name}_add_batch")(ptr(acc), ptr(pts[i:i + 1].copy()), ptr(nxt), 1) == 0
            acc = nxt
        assert (got == acc).all(), (name, cnt)
    # the two orders the fold must not depend on: (P0 + P0') first and last — same group element, canonical output
    perm = pts[::-1].copy(); got2 = np.zeros((1, W), np.uint64)
    zk.check(getattr(L, f"zkt_{name}_sum")(perm.ctypes.data, n, got2.ctypes.data))
    assert (got2 == got).all()
    for k in (rng.below(order), 0, order, (1 << 256) - 1):
        kk = ints_to_arr([k], 4)
        got, want = np.zeros_like(pts), np.zeros_like(pts)
        zk.check(getattr(L, f"zkt_{name}_scale_batch")(pts.ctypes.data, kk.ctypes.data, 4, got.ctypes.data, n))
        assert getattr(O, f"zkto_{name}_mul_batch")(ptr(pts), ptr(np.repeat(kk, n, axis=0)), 4, ptr(want), n, 8) == 0
        assert (got == want).all(), (name, k)
    zk.check(getattr(L, f"zkt_{name}_scale_batch")(pts.ctypes.data, kk.ctypes.data, 4, got.ctypes.data, 0))      # empty vector: allowed (no assert in the reference)


def test_tate_dev_from_many_caller_streams(L):
    """zkt_tate_batch_dev driven from EIGHT streams of the caller in one process: the pairing kernels keep 10-17 KB of scratch per
    lane, which the runtime retains per hardware queue (5-6 GiB each) — so the library runs them on its own staging stream, ordered behind the caller's
    stream by an event, whatever stream the caller names.  All eight results (inputs produced on the caller's stream just before) equal the host-pointer
    call; device memory retained afterwards stays far below eight queues' worth of scratch."""
    import torch
    n = 256
    rng = SplitMix64(9090)
    g1 = np.repeat(_gen(0), n, axis=0); g2 = np.repeat(_gen(1), n, axis=0)
    kp = ints_to_arr([rng.below(R) for _ in range(n)], 4); kq = ints_to_arr([rng.below(R) for _ in range(n)], 4)
    P, Qs = np.zeros_like(g1), np.zeros_like(g2)
    zk.check(L.zkt_g1_mul_batch(ptr(g1), ptr(kp), 4, ptr(P), n)); zk.check(L.zkt_g2_mul_batch(ptr(g2), ptr(kq), 4, ptr(Qs), n))
    want = np.zeros((n, FQ12), np.uint64); zk.check(L.zkt_tate_batch(ptr(P), ptr(Qs), ptr(want), n))
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    free0 = torch.cuda.mem_get_info()[0]
    streams = [torch.cuda.Stream() for _ in range(8)]
    outs = []
    for k, st in enumerate(streams):
        with torch.cuda.stream(st):
            # the inputs are produced ON the caller's stream (a copy + a roll that puts them back in place), so the ordering event matters
            d_p = torch.roll(torch.roll(torch.from_numpy(P.view(np.int64)).cuda(non_blocking=True), k, 0), -k, 0)
            d_q = torch.roll(torch.roll(torch.from_numpy(Qs.view(np.int64)).cuda(non_blocking=True), k, 0), -k, 0)
            d_e = torch.empty((n, FQ12), dtype=torch.int64, device="cuda")
            zk.check(L.zkt_tate_batch_dev(vp(d_p), vp(d_q), vp(d_e), n, ctypes.c_void_p(st.cuda_stream)))
            outs.append(d_e)
    torch.cuda.synchronize()
    for k, d_e in enumerate(outs):
        assert (d_e.cpu().numpy().view(np.uint64) == want).all(), f"stream {k}"
    retained = free0 - torch.cuda.mem_get_info()[0]
    assert retained < 8 * (1 << 30), f"{retained / 2**30:.1f} GiB retained after eight caller streams: the pairing kernels ran on the callers' queues"
