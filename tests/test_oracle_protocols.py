"""Oracle restatement of the two callers (Groth16, Bulletproofs IPA) checked through the reference's own
acceptance tests: a generated proof verifies (prover.rs:159-192), the IPA accepts an honest opening
(bulletproofs.rs:231-282 shape).  CPU only."""
import ctypes
import numpy as np
from zkt_testlib import *
from qap_util import *

O = oracle()
fr = lambda v: ints_to_arr([v], 4)


def groth16_oracle(A, B, C, wit, l, seed, literal):
    n, m = len(A), len(wit) - 1
    ui, vi, wi, h, _ = qap_from_r1cs(A, B, C, wit)
    U, V, W = dense(ui, n), dense(vi, n), dense(wi, n)
    rng = SplitMix64(seed)
    trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
    crs, bufs = alloc_crs(n, l, m)
    assert O.zkto_groth16_setup(ctypes.byref(crs), ptr(U), ptr(V), ptr(W), *[ptr(t) for t in trap]) == 0
    wires = ints_to_arr(wit, 4); H = ints_to_arr(h, 4)
    r, s = fr(rng.below(R - 1) + 1), fr(rng.below(R - 1) + 1)
    pa, pb, pc = np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64)
    assert O.zkto_groth16_prove(ctypes.byref(crs), ptr(U), ptr(V), ptr(wires), ptr(H), len(h), ptr(r), ptr(s), literal, ptr(pa), ptr(pb), ptr(pc)) == 0
    return crs, bufs, (U, V, W, wires, H, len(h), trap, r, s), (pa, pb, pc)


def test_groth16_reference_example_verifies():          # prover.rs:159-192: (x*x*x)+x+5==35, x=3
    A, B, C, wit, l = example_cubic()
    crs, bufs, inp, (pa, pb, pc) = groth16_oracle(A, B, C, wit, l, 7, literal=1)
    stmt = ints_to_arr(wit[:l + 1], 4)
    assert O.zkto_groth16_verify(ctypes.byref(crs), ptr(pa), ptr(pb), ptr(pc), ptr(stmt), l + 1) == 1
    # the one-MSM-per-output form gives the same proof points as the reference's per-wire loop
    _, _, _, (qa, qb, qc) = groth16_oracle(A, B, C, wit, l, 7, literal=0)
    assert (pa == qa).all() and (pb == qb).all() and (pc == qc).all()
    # a wrong statement is rejected
    bad = ints_to_arr([1, 3, 36], 4)
    assert O.zkto_groth16_verify(ctypes.byref(crs), ptr(pa), ptr(pb), ptr(pc), ptr(bad), l + 1) == 0


def ipa_instance(n, seed):
    rng = SplitMix64(seed)
    g = np.zeros((1, 9), np.uint64); O.zkto_secp_generator(ptr(g))
    ks = ints_to_arr([rng.below(SECP_N - 1) + 1 for _ in range(2 * n + 1)], 4)
    pts = np.zeros((2 * n + 1, 9), np.uint64)
    assert O.zkto_secp_mul_batch(ptr(np.repeat(g, 2 * n + 1, axis=0)), ptr(ks), 4, ptr(pts), 2 * n + 1, 8) == 0
    gg, hh, u = pts[:n].copy(), pts[n:2 * n].copy(), pts[2 * n:].copy()
    a = ints_to_arr([rng.below(SECP_N) for _ in range(n)], 4); b = ints_to_arr([rng.below(SECP_N) for _ in range(n)], 4)
    P = np.zeros((1, 9), np.uint64)
    assert O.zkto_bp_commit(n, ptr(gg), ptr(hh), ptr(u), ptr(a), ptr(b), ptr(P)) == 0
    levels = max(n.bit_length() - 1, 1)
    xs = ints_to_arr([rng.below(SECP_N - 1) + 1 for _ in range(levels)], 4)
    return gg, hh, u, P, a, b, xs


def test_ipa_accepts_honest_and_rejects_tampered():    # bulletproofs.rs:231-246 (n=2, gg=[g,g]) and the recursive case
    for n in (1, 2, 4, 8):
        gg, hh, u, P, a, b, xs = ipa_instance(n, 90 + n)
        assert O.zkto_bp_ipa(n, ptr(gg), ptr(hh), ptr(u), ptr(P), ptr(a), ptr(b), ptr(xs), None) == 1
        b2 = b.copy(); b2[0, 0] ^= 1
        assert O.zkto_bp_ipa(n, ptr(gg), ptr(hh), ptr(u), ptr(P), ptr(a), ptr(b2), ptr(xs), None) == 0
    # repeated generators gg = [g, g] as in the reference's test
    gg, hh, u, P, a, b, xs = ipa_instance(2, 5)
    gg[1] = gg[0]
    assert O.zkto_bp_commit(2, ptr(gg), ptr(hh), ptr(u), ptr(a), ptr(b), ptr(P)) == 0
    assert O.zkto_bp_ipa(2, ptr(gg), ptr(hh), ptr(u), ptr(P), ptr(a), ptr(b), ptr(xs), None) == 1


def range_proof_instance(n, value, seed):
    rng = SplitMix64(seed)
    g0 = np.zeros((1, 9), np.uint64); O.zkto_secp_generator(ptr(g0))
    ks = ints_to_arr([rng.below(SECP_N - 1) + 1 for _ in range(2 * n + 3)], 4)
    pts = np.zeros((2 * n + 3, 9), np.uint64)
    assert O.zkto_secp_mul_batch(ptr(np.repeat(g0, 2 * n + 3, axis=0)), ptr(ks), 4, ptr(pts), 2 * n + 3, 8) == 0
    gg, hh, g, h, u = pts[:n].copy(), pts[n:2 * n].copy(), pts[2 * n:2 * n + 1].copy(), pts[2 * n + 1:2 * n + 2].copy(), pts[2 * n + 2:].copy()
    aL = ints_to_arr([(value >> i) & 1 for i in range(n)], 4)            # bits of the value (bulletproofs.rs:258-262)
    gamma = ints_to_arr([rng.below(SECP_N - 1) + 1], 4)
    V = np.zeros((1, 9), np.uint64)                                         # V = g^v h^gamma
    tmp = np.zeros((2, 9), np.uint64)
    assert O.zkto_secp_mul_batch(ptr(np.concatenate([g, h])), ptr(np.concatenate([ints_to_arr([value], 4), gamma])), 4, ptr(tmp), 2, 1) == 0
    assert O.zkto_secp_add_batch(ptr(tmp[0:1].copy()), ptr(tmp[1:2].copy()), ptr(V), 1) == 0
    rnd = ints_to_arr([rng.below(SECP_N - 1) + 1 for _ in range(7 + 2 * n)], 4)
    xs = ints_to_arr([rng.below(SECP_N - 1) + 1 for _ in range(max(n.bit_length() - 1, 1))], 4)
    return V, aL, gamma, g, h, gg, hh, rnd, u, xs


def test_range_proof_accepts_in_range_value():      # bulletproofs.rs:248-282: n = 4, value 9, with and without the IPA
    for use_ipa in (0, 1):
        V, aL, gamma, g, h, gg, hh, rnd, u, xs = range_proof_instance(4, 9, 31)
        assert O.zkto_bp_range_proof(4, ptr(V), ptr(aL), ptr(gamma), ptr(g), ptr(h), ptr(gg), ptr(hh), use_ipa, ptr(rnd), ptr(u), ptr(xs), None) == 1
        bad = aL.copy(); bad[0, 0] ^= 1                                    # opening that does not match V
        assert O.zkto_bp_range_proof(4, ptr(V), ptr(bad), ptr(gamma), ptr(g), ptr(h), ptr(gg), ptr(hh), use_ipa, ptr(rnd), ptr(u), ptr(xs), None) == 0
