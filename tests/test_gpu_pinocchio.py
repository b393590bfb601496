"""Pinocchio setup / prove / verify on the GPU against the oracle's line-by-line restatement of
pinocchio/{crs.rs:49-161, prover.rs:98-170, verifier.rs:31-85} with the same injected randomness: every CRS element and every
proof point must be bit-identical; accept / reject / panic decisions must agree."""
import ctypes, importlib
import numpy as np
import pytest
from zkt_testlib import *
from qap_util import *

pytestmark = pytest.mark.gpu
zk = importlib.import_module("zk-toolkit_amd")
O = oracle()


@pytest.fixture(scope="module")
def L():
    zk.init()
    return zk.lib()


@pytest.mark.parametrize("case", ["cubic", "chain4", "chain9"])
def test_pinocchio_vs_oracle(L, case):
    A, B, C, wit, l = example_cubic() if case == "cubic" else chain_circuit(int(case[5:]))      # cubic: the reference's own test, prover.rs:177-211
    n, n_io = len(A), l + 1
    n_mid = len(wit) - n_io
    V, W, Y, h, max_degree = pinocchio_instance(A, B, C, wit)
    rng = SplitMix64(77 + n)
    rnd = ints_to_arr([rng.below(R - 1) + 1 for _ in range(8)], 4)
    dv, dy = ints_to_arr([rng.below(R - 1) + 1], 4), ints_to_arr([rng.below(R - 1) + 1], 4)
    wires, H = ints_to_arr(wit, 4), ints_to_arr(h, 4)
    ocrs, obuf = alloc_pinocchio(n, n_io, n_mid, max_degree); gcrs, gbuf = alloc_pinocchio(n, n_io, n_mid, max_degree)
    assert O.zkto_pinocchio_setup(ctypes.byref(ocrs), ptr(V), ptr(W), ptr(Y), ptr(rnd)) == 0
    zk.check(L.zkt_pinocchio_setup(ctypes.byref(gcrs), ptr(V), ptr(W), ptr(Y), ptr(rnd)))
    for k in obuf:
        assert (obuf[k] == gbuf[k]).all(), f"CRS field {k} differs"
    opf, opb = alloc_pinocchio_proof(); gpf, gpb = alloc_pinocchio_proof()
    assert O.zkto_pinocchio_prove(ctypes.byref(ocrs), ptr(wires), ptr(H), len(h), ptr(dv), ptr(dy), ctypes.byref(opf)) == 0
    zk.check(L.zkt_pinocchio_prove(ctypes.byref(gcrs), ptr(wires), ptr(H), len(h), ptr(dv), ptr(dy), ctypes.byref(gpf)))
    for k in opb:
        if not (opb[k] == gpb[k]).all():             # say whether a second evaluation agrees with the oracle (a transient would point at a race, not at the arithmetic)
            gpf2, gpb2 = alloc_pinocchio_proof()
            zk.check(L.zkt_pinocchio_prove(ctypes.byref(gcrs), ptr(wires), ptr(H), len(h), ptr(dv), ptr(dy), ctypes.byref(gpf2)))
            raise AssertionError(f"proof element {k} differs from the oracle; a second GPU evaluation {'agrees' if (opb[k] == gpb2[k]).all() else 'differs too'}; "
                                 f"first differing word {int(np.argmax((opb[k] != gpb[k]).ravel()))} of {opb[k].size}")
    io = wires[:n_io].copy()
    both = lambda pf_o, pf_g, w: (O.zkto_pinocchio_verify(ctypes.byref(ocrs), ctypes.byref(pf_o), ptr(w)), L.zkt_pinocchio_verify(ctypes.byref(gcrs), ctypes.byref(pf_g), ptr(w)))
    assert both(opf, gpf, io) == (1, 1)
    bad = io.copy(); bad[n_io - 1, 0] ^= np.uint64(1)
    assert both(opf, gpf, bad) == (0, 0)                                                         # divisibility check fails
    for buf in (opb, gpb): buf["alpha_y_mid_s"][:] = buf["alpha_v_mid_s"]
    assert both(opf, gpf, io) == (0, 0)                                                          # knowledge-of-coefficient check of y fails
    # an argument at infinity in the third check; the second check still passes, so the reference panics there
    for buf in (opb, gpb): buf["alpha_y_mid_s"][:] = opb["alpha_v_mid_s"]; buf["g2_w_mid_s"][:] = 0; buf["g2_w_mid_s"][0, 24] = 1
    assert both(opf, gpf, io) == (-2, -ZKT_ERR_INFINITY)
    # ... but a rejection by an earlier check wins over the later panic (verifier.rs returns before evaluating it)
    for buf in (opb, gpb): buf["alpha_v_mid_s"][:] = buf["v_mid_s"]
    assert both(opf, gpf, io) == (0, 0)


def _instance(L, constraints, seed):
    A, B, C, wit, l = chain_circuit(constraints)
    n, n_io = len(A), l + 1
    V, W, Y, h, max_degree = pinocchio_instance(A, B, C, wit)
    rng = SplitMix64(seed)
    rnd = ints_to_arr([rng.below(R - 1) + 1 for _ in range(8)], 4)
    dv, dy = ints_to_arr([rng.below(R - 1) + 1], 4), ints_to_arr([rng.below(R - 1) + 1], 4)
    wires, H = ints_to_arr(wit, 4), ints_to_arr(h, 4)
    crs, cbuf = alloc_pinocchio(n, n_io, len(wit) - n_io, max_degree)
    zk.check(L.zkt_pinocchio_setup(ctypes.byref(crs), ptr(V), ptr(W), ptr(Y), ptr(rnd)))
    pf, pbuf = alloc_pinocchio_proof()
    zk.check(L.zkt_pinocchio_prove(ctypes.byref(crs), ptr(wires), ptr(H), len(h), ptr(dv), ptr(dy), ctypes.byref(pf)))
    return crs, cbuf, pf, pbuf, wires[:n_io].copy()


def test_pinocchio_verify_with_several_keys(L):
    """The verifier keeps fixed-base tables of the last two keys' io points (zkt_pinocchio.hip): alternate between three keys so that entries are
    reused, evicted and rebuilt, and compare every decision with the oracle's verifier on the same key, proof and statement."""
    inst = [_instance(L, 4, 501), _instance(L, 4, 502), _instance(L, 6, 503)]
    def both(k, io):
        crs, _, pf, _, _ = inst[k]
        return O.zkto_pinocchio_verify(ctypes.byref(crs), ctypes.byref(pf), ptr(io)), L.zkt_pinocchio_verify(ctypes.byref(crs), ctypes.byref(pf), ptr(io))
    for k in (0, 1, 0, 1, 2, 0, 2, 1):
        io = inst[k][4]
        assert both(k, io) == (1, 1)
        bad = io.copy(); bad[1, 1] ^= np.uint64(5)
        assert both(k, bad) == (0, 0)
    # a proof checked against another key of the same shape
    crs1, pf0, io0 = inst[1][0], inst[0][2], inst[0][4]
    assert O.zkto_pinocchio_verify(ctypes.byref(crs1), ctypes.byref(pf0), ptr(io0)) == 0
    assert L.zkt_pinocchio_verify(ctypes.byref(crs1), ctypes.byref(pf0), ptr(io0)) == 0
    # statement wires of zero: every product is the point at infinity and the sums are the proof's own points
    zero = np.zeros_like(inst[0][4])
    assert both(0, zero) == (0, 0)
    # only the last equality has an argument at infinity (h_s): the first four pass, the reference panics in the fifth
    inst[2][3]["h_s"][:] = 0; inst[2][3]["h_s"][0, 24] = 1
    assert both(2, inst[2][4]) == (-2, -ZKT_ERR_INFINITY)
    # ... and a statement that is larger than the group order is used as the integer it is, on both sides
    big = inst[1][4].copy(); big[1] = np.array(int_to_limbs(int(limbs_to_int(inst[1][4][1])) + R, 4), np.uint64)
    assert both(1, big) == (1, 1)


def test_pinocchio_verify_points_outside_g2(L):
    """The fused five-equality launch runs the 127-step Miller loop, which needs every G2 argument in G2; a proof carrying a point of the twist outside G2
    (or off the twist) must come out as the reference's own evaluation decides — the verifier falls back to the table-free path for that proof."""
    crs, cbuf, pf, pbuf, io = _instance(L, 5, 777)
    both = lambda: (O.zkto_pinocchio_verify(ctypes.byref(crs), ctypes.byref(pf), ptr(io)), L.zkt_pinocchio_verify(ctypes.byref(crs), ctypes.byref(pf), ptr(io)))
    assert both() == (1, 1)
    rng = SplitMix64(99)
    keep = {k: pbuf[k].copy() for k in ("h_s", "g2_w_mid_s")}
    pbuf["h_s"][:] = g2_arr([to_abi_g2(py_twist_point(rng))])                                 # on E', outside G2: only the divisibility check sees it
    o, g = both(); assert o == g and o in (0, 1)
    pbuf["h_s"][:] = keep["h_s"]
    pbuf["g2_w_mid_s"][:] = g2_arr([to_abi_g2(py_twist_point(rng))])                          # used by the alpha_w check and, through w_s, by the last one
    o, g = both(); assert o == g and o in (0, 1)
    (x1, x0), (y1, y0) = g2_from_arr(keep["g2_w_mid_s"])[0]
    pbuf["g2_w_mid_s"][:] = g2_arr([((x1, x0), (y1, (y0 + 1) % Q))])                           # off the twist: the reference adds and pairs it all the same, or panics
    o, g = both(); assert (o, g) in ((0, 0), (1, 1)) or (o < 0 and g < 0)
    pbuf["g2_w_mid_s"][:] = keep["g2_w_mid_s"]
    assert both() == (1, 1)


def test_pinocchio_verify_g1_points_outside_the_subgroup(L):
    """A proof carrying a G1 element on the curve outside the order-r subgroup: the reference pairs it all the same (a value that depends on the point's order) or panics
    inside tate (rational_function.rs:36); verifier.rs:43-84 then accepts, rejects or unwinds accordingly.  The engine must reach the same outcome — such elements leave
    the product form and are evaluated side by side through the reference's chain (zkt_pairing.hip, k_product_exact_marked)."""
    crs, cbuf, pf, pbuf, io = _instance(L, 5, 778)
    both = lambda: (O.zkto_pinocchio_verify(ctypes.byref(crs), ctypes.byref(pf), ptr(io)), L.zkt_pinocchio_verify(ctypes.byref(crs), ctypes.byref(pf), ptr(io)))
    assert both() == (1, 1)
    seen = set()
    for field in ("alpha_v_mid_s", "y_mid_s", "beta_vwy_mid_s"):
        keep = pbuf[field].copy()
        for label, pt in degenerate_g1_points():
            pbuf[field][:] = g1_arr([pt])
            o, g = both()
            assert (o, g) in ((0, 0), (1, 1)) or (o < 0 and g < 0), (field, label, o, g)
            seen.add(o if o >= 0 else -1)
        pbuf[field][:] = keep
    assert both() == (1, 1) and 0 in seen


@pytest.mark.parametrize("case", ["cubic", "chain9"])
def test_pinocchio_resident_prover_equals_one_shot(L, case):
    """zkt_pinocchio_prove_resident (evaluation key in HBM, wires uploaded once, ten pipelined MSMs) gives the nine proof points of zkt_pinocchio_prove, which the
    test above pins to the oracle's prover.rs:98-170 — twice on one handle, and the proof verifies."""
    A, B, C, wit, l = example_cubic() if case == "cubic" else chain_circuit(int(case[5:]))
    n, n_io = len(A), l + 1
    V, W, Y, h, max_degree = pinocchio_instance(A, B, C, wit)
    rng = SplitMix64(177 + n)
    rnd = ints_to_arr([rng.below(R - 1) + 1 for _ in range(8)], 4)
    wires, H = ints_to_arr(wit, 4), ints_to_arr(h, 4)
    crs, cbuf = alloc_pinocchio(n, n_io, len(wit) - n_io, max_degree)
    zk.check(L.zkt_pinocchio_setup(ctypes.byref(crs), ptr(V), ptr(W), ptr(Y), ptr(rnd)))
    pk = ctypes.c_void_p()
    zk.check(L.zkt_pinocchio_pk_create(ctypes.byref(crs), ctypes.byref(pk)))
    try:
        for rep in range(2):
            dv, dy = ints_to_arr([rng.below(R - 1) + 1], 4), ints_to_arr([rng.below(R - 1) + 1], 4)
            pf1, b1 = alloc_pinocchio_proof(); pf2, b2 = alloc_pinocchio_proof()
            zk.check(L.zkt_pinocchio_prove(ctypes.byref(crs), ptr(wires), ptr(H), len(h), ptr(dv), ptr(dy), ctypes.byref(pf1)))
            zk.check(L.zkt_pinocchio_prove_resident(pk, ptr(wires), ptr(H), len(h), ptr(dv), ptr(dy), ctypes.byref(pf2)))
            for k in b1:
                assert (b1[k] == b2[k]).all(), (rep, k)
            assert L.zkt_pinocchio_verify(ctypes.byref(crs), ctypes.byref(pf2), ptr(wires[:n_io].copy())) == 1
        assert L.zkt_pinocchio_prove_resident(pk, ptr(wires), ptr(H), max_degree + 1, ptr(dv), ptr(dy), ctypes.byref(pf2)) == ZKT_ERR_SHAPE      # polynomial.rs:289-291
    finally:
        L.zkt_pinocchio_pk_free(pk)


def test_pinocchio_resident_prover_2p16_wires_by_linearity(L):
    """The prover at the size the protocol runs (2^16 mid wires, quotient of degree 2^16 - 1): an evaluation key whose every base is a KNOWN multiple of the
    generator, so each proof point is generator * (a sum computed here with python integers) — prover.rs:124-161 restated on the exponents."""
    n_mid, n_io, deg = 1 << 16, 3, 1 << 16
    gen = np.random.Generator(np.random.PCG64(31))
    def scal(cnt):
        a = gen.integers(0, 2**63, size=(cnt, 4), dtype=np.uint64); a[:, 3] >>= np.uint64(2); return a
    g1 = np.zeros((1, G1W), np.uint64); O.zkto_g1_generator(ptr(g1)); g2 = np.zeros((1, G2W), np.uint64); O.zkto_g2_generator(ptr(g2))
    crs, buf = alloc_pinocchio(n_mid, n_io, n_mid, deg)
    exps = {}
    for name, w, c in [e for e in (qap_util_pin_ek() + qap_util_pin_vk())]:
        cnt = buf[name].shape[0]
        k = scal(cnt); exps[name] = [int(limbs_to_int(r)) % R for r in k]
        if w == G1W: zk.check(L.zkt_bls_public_keys_batch(ptr(k), cnt, ptr(buf[name])))            # generator * k through the comb table
        else: zk.check(L.zkt_g2_mul_batch(ptr(np.repeat(g2, cnt, axis=0)), ptr(k), 4, ptr(buf[name]), cnt))
    wires = scal(n_io + n_mid); wires[5] = 0; wires[6] = np.array([1, 0, 0, 0], np.uint64)
    H = scal(deg)
    dv, dy = scal(1), scal(1)
    wv = [int(limbs_to_int(r)) for r in wires]; hv = [int(limbs_to_int(r)) for r in H]
    dvi, dyi = int(limbs_to_int(dv[0])), int(limbs_to_int(dy[0]))
    mid = wv[n_io:]
    dot = lambda name, vals: sum(a * b for a, b in zip(exps[name], vals)) % R
    t, avt, ayt, bt, one2 = exps["t"][0], exps["alpha_v_t"][0], exps["alpha_y_t"][0], exps["beta_t"][0], exps["one_g2"][0]
    w_s = (dot("g2_wk_mid", mid) + dot("wk_io", wv[:n_io])) % R
    want = {"v_mid_s": (t * dvi + dot("vk_mid", mid)) % R, "g1_w_mid_s": dot("g1_wk_mid", mid), "g2_w_mid_s": dot("g2_wk_mid", mid), "y_mid_s": (t * dyi + dot("yk_mid", mid)) % R,
            "alpha_v_mid_s": (avt * dvi + dot("alpha_vk_mid", mid)) % R, "alpha_w_mid_s": dot("alpha_wk_mid", mid), "alpha_y_mid_s": (ayt * dyi + dot("alpha_yk_mid", mid)) % R,
            "beta_vwy_mid_s": (bt * dvi + bt * dyi + dot("beta_vwy_k_mid", mid)) % R, "h_s": (dot("si", hv) + w_s * dvi - one2 * dyi) % R}
    pk = ctypes.c_void_p()
    zk.check(L.zkt_pinocchio_pk_create(ctypes.byref(crs), ctypes.byref(pk)))
    try:
        pf, pb = alloc_pinocchio_proof()
        zk.check(L.zkt_pinocchio_prove_resident(pk, ptr(wires), ptr(H), deg, ptr(dv), ptr(dy), ctypes.byref(pf)))
        for name, e in want.items():
            exp = np.zeros((1, pb[name].shape[1]), np.uint64)
            if pb[name].shape[1] == G1W: assert O.zkto_g1_mul_batch(ptr(g1), ptr(ints_to_arr([e], 4)), 4, ptr(exp), 1, 1) == 0
            else: assert O.zkto_g2_mul_batch(ptr(g2), ptr(ints_to_arr([e], 4)), 4, ptr(exp), 1, 1) == 0
            assert (pb[name] == exp).all(), name
    finally:
        L.zkt_pinocchio_pk_free(pk)
