"""GPU parity for the two callers of the hot path against the oracle's
restatement with the same injected randomness: CRS points, proof points (A,B,C) and IPA transcripts must be
bit-identical; accept/reject decisions must agree."""
import ctypes, importlib
import numpy as np
import pytest
from zkt_testlib import *
from qap_util import *

pytestmark = pytest.mark.gpu
zk = importlib.import_module("zk-toolkit_amd")
O = oracle()
fr = lambda v: ints_to_arr([v], 4)


@pytest.fixture(scope="module")
def L():
    zk.init()
    return zk.lib()


@pytest.mark.parametrize("n", [1, 3, 16, 100])
def test_g2_and_secp_msm_vs_oracle(L, n):           # polynomial.rs:283-293; secp256k1/affine_points.rs:25-31,123-144
    rng = SplitMix64(500 + n)
    for name, W, order, gen in (("g2", G2W, R, O.zkto_g2_generator), ("secp", 9, SECP_N, O.zkto_secp_generator)):
        g = np.zeros((1, W), np.uint64); gen(ptr(g))
        bases = np.zeros((n, W), np.uint64)
        zk.check(getattr(L, f"zkt_{name}_mul_batch")(ptr(np.repeat(g, n, axis=0)), ptr(ints_to_arr([rng.below(order) for _ in range(n)], 4)), 4, ptr(bases), n))
        ss = [rng.below(order) for _ in range(n)]
        if n > 2: ss[1] = 0
        sc = ints_to_arr(ss, 4)
        got = np.zeros((1, W), np.uint64); zk.check(getattr(L, f"zkt_{name}_msm")(ptr(bases), ptr(sc), n, ptr(got)))
        # oracle: n scalar muls then sequential adds
        tmp = np.zeros_like(bases); assert getattr(O, f"zkto_{name}_mul_batch")(ptr(bases), ptr(sc), 4, ptr(tmp), n, 8) == 0
        acc = np.zeros((1, W), np.uint64); acc[0, W - 1] = 1
        for i in range(n):
            nxt = np.zeros_like(acc); assert getattr(O, f"zkto_{name}_add_batch")(ptr(acc), ptr(tmp[i:i + 1].copy()), ptr(nxt), 1) == 0; acc = nxt
        assert (got == acc).all(), (name, n)


def run_both(L, A, B, C, wit, l, seed, literal):
    n, m = len(A), len(wit) - 1
    ui, vi, wi, h, _ = qap_from_r1cs(A, B, C, wit)
    U, V, W = dense(ui, n), dense(vi, n), dense(wi, n)
    rng = SplitMix64(seed)
    trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
    r, s = fr(rng.below(R - 1) + 1), fr(rng.below(R - 1) + 1)
    wires = ints_to_arr(wit, 4); H = ints_to_arr(h, 4)
    ocrs, obuf = alloc_crs(n, l, m); gcrs, gbuf = alloc_crs(n, l, m)
    assert O.zkto_groth16_setup(ctypes.byref(ocrs), ptr(U), ptr(V), ptr(W), *[ptr(t) for t in trap]) == 0
    zk.check(L.zkt_groth16_setup(ctypes.byref(gcrs), ptr(U), ptr(V), ptr(W), *[ptr(t) for t in trap]))
    for k in obuf:
        assert (obuf[k] == gbuf[k]).all(), f"CRS field {k} differs"
    op = (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
    gp = (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
    assert O.zkto_groth16_prove(ctypes.byref(ocrs), ptr(U), ptr(V), ptr(wires), ptr(H), len(h), ptr(r), ptr(s), literal, *[ptr(x) for x in op]) == 0
    zk.check(L.zkt_groth16_prove(ctypes.byref(gcrs), ptr(U), ptr(V), ptr(wires), ptr(H), len(h), ptr(r), ptr(s), *[ptr(x) for x in gp]))
    for a, b, name in zip(op, gp, "ABC"):
        assert (a == b).all(), f"proof element {name} differs"
    stmt = ints_to_arr(wit[:l + 1], 4)
    assert L.zkt_groth16_verify(ctypes.byref(gcrs), *[ptr(x) for x in gp], ptr(stmt), l + 1) == 1
    assert O.zkto_groth16_verify(ctypes.byref(ocrs), *[ptr(x) for x in gp], ptr(stmt), l + 1) == 1
    bad = stmt.copy(); bad[l, 0] ^= 1
    assert L.zkt_groth16_verify(ctypes.byref(gcrs), *[ptr(x) for x in gp], ptr(bad), l + 1) == 0
    tam = gp[2].copy(); tam[0, :12] = gp[0][0, :12]                      # C := A
    assert L.zkt_groth16_verify(ctypes.byref(gcrs), ptr(gp[0]), ptr(gp[1]), ptr(tam), ptr(stmt), l + 1) == 0


def test_groth16_reference_example(L):              # prover.rs:159-192, literal per-wire loop in the oracle
    run_both(L, *example_cubic(), seed=11, literal=1)


@pytest.mark.parametrize("n", [16, 64])
def test_groth16_chain_circuit(L, n):               # SURVEY §8d C4 generator at parity sizes
    run_both(L, *chain_circuit(n), seed=12 + n, literal=0)


def test_groth16_verify_infinity_is_error(L):       # verifier.rs:40-49: tate with the point at infinity panics
    A, B, C, wit, l = example_cubic()
    n, m = len(A), len(wit) - 1
    ui, vi, wi, h, _ = qap_from_r1cs(A, B, C, wit)
    crs, buf = alloc_crs(n, l, m)
    rng = SplitMix64(3); trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
    zk.check(L.zkt_groth16_setup(ctypes.byref(crs), ptr(dense(ui, n)), ptr(dense(vi, n)), ptr(dense(wi, n)), *[ptr(t) for t in trap]))
    inf1 = g1_arr([None]); g2 = buf["g2_beta"]
    assert L.zkt_groth16_verify(ctypes.byref(crs), ptr(inf1), ptr(g2), ptr(buf["g1_alpha"]), ptr(ints_to_arr(wit[:l + 1], 4)), l + 1) == -ZKT_ERR_INFINITY


def ipa_instance(L, n, seed):
    rng = SplitMix64(seed)
    g = np.zeros((1, 9), np.uint64); O.zkto_secp_generator(ptr(g))
    ks = ints_to_arr([rng.below(SECP_N - 1) + 1 for _ in range(2 * n + 1)], 4)
    pts = np.zeros((2 * n + 1, 9), np.uint64)
    zk.check(L.zkt_secp_mul_batch(ptr(np.repeat(g, 2 * n + 1, axis=0)), ptr(ks), 4, ptr(pts), 2 * n + 1))
    gg, hh, u = pts[:n].copy(), pts[n:2 * n].copy(), pts[2 * n:].copy()
    av = [rng.below(SECP_N) for _ in range(n)]; bv = [rng.below(SECP_N) for _ in range(n)]
    a, b = ints_to_arr(av, 4), ints_to_arr(bv, 4)
    # P = g^a h^b u^<a,b> (bulletproofs.rs:17) through the GPU MSM
    c = sum(x * y for x, y in zip(av, bv)) % SECP_N
    P = np.zeros((1, 9), np.uint64)
    zk.check(L.zkt_secp_msm(ptr(np.concatenate([gg, hh, u])), ptr(np.concatenate([a, b, ints_to_arr([c], 4)])), 2 * n + 1, ptr(P)))
    levels = max(n.bit_length() - 1, 1)
    xs = ints_to_arr([rng.below(SECP_N - 1) + 1 for _ in range(levels)], 4)
    return gg, hh, u, P, a, b, xs, levels


@pytest.mark.parametrize("n", [1, 2, 4, 64])
def test_ipa_transcript_vs_oracle(L, n):            # bulletproofs.rs:19-55, level by level
    gg, hh, u, P, a, b, xs, levels = ipa_instance(L, n, 600 + n)
    want_P = np.zeros((1, 9), np.uint64); assert O.zkto_bp_commit(n, ptr(gg), ptr(hh), ptr(u), ptr(a), ptr(b), ptr(want_P)) == 0
    assert (P == want_P).all()
    gt, ot = np.zeros((levels * 3, 9), np.uint64), np.zeros((levels * 3, 9), np.uint64)
    assert L.zkt_bp_inner_product_argument(n, ptr(gg), ptr(hh), ptr(u), ptr(P), ptr(a), ptr(b), ptr(xs), ptr(gt)) == 1
    assert O.zkto_bp_ipa(n, ptr(gg), ptr(hh), ptr(u), ptr(P), ptr(a), ptr(b), ptr(xs), ptr(ot)) == 1
    if n > 1: assert (gt == ot).all()
    b2 = b.copy(); b2[0, 0] ^= 1
    assert L.zkt_bp_inner_product_argument(n, ptr(gg), ptr(hh), ptr(u), ptr(P), ptr(a), ptr(b2), ptr(xs), None) == 0


@pytest.mark.parametrize("n", [1, 8, 256])
def test_ipa_resident_generators(L, n):             # zkt_bp_ipa_ctx: one context, several arguments, same transcript as the one-shot form and the oracle
    gg, hh, u, P, a, b, xs, levels = ipa_instance(L, n, 900 + n)
    ctx = ctypes.c_void_p()
    zk.check(L.zkt_bp_ipa_ctx_create(n, ptr(gg), ptr(hh), ptr(u), ctypes.byref(ctx)))
    try:
        gt, ot = np.zeros((levels * 3, 9), np.uint64), np.zeros((levels * 3, 9), np.uint64)
        assert O.zkto_bp_ipa(n, ptr(gg), ptr(hh), ptr(u), ptr(P), ptr(a), ptr(b), ptr(xs), ptr(ot)) == 1
        for _ in range(2):                            # the context is reusable
            gt[:] = 0
            assert L.zkt_bp_inner_product_argument_ctx(ctx, ptr(P), ptr(a), ptr(b), ptr(xs), ptr(gt)) == 1
            if n > 1: assert (gt == ot).all()
        assert L.zkt_bp_inner_product_argument_ctx(ctx, ptr(P), ptr(a), ptr(b), ptr(xs), None) == 1           # no-trace path (block sum of the products)
        b2 = b.copy(); b2[n - 1, 1] ^= 4
        assert L.zkt_bp_inner_product_argument_ctx(ctx, ptr(P), ptr(a), ptr(b2), ptr(xs), None) == 0
        if n > 1:
            x0 = xs.copy(); x0[levels - 1] = 0        # a non-invertible challenge is refused and leaves the context usable
            assert L.zkt_bp_inner_product_argument_ctx(ctx, ptr(P), ptr(a), ptr(b), ptr(x0), None) < 0
        assert L.zkt_bp_inner_product_argument_ctx(ctx, ptr(P), ptr(a), ptr(b), ptr(xs), None) == 1
    finally:
        L.zkt_bp_ipa_ctx_free(ctx)


def test_ipa_full_size_accepts(L):                  # BASELINE config 5 shape: n = 64*1024 generators, 16 levels
    n = 1 << 16
    gg, hh, u, P, a, b, xs, levels = ipa_instance(L, n, 77)
    assert levels == 16
    assert L.zkt_bp_inner_product_argument(n, ptr(gg), ptr(hh), ptr(u), ptr(P), ptr(a), ptr(b), ptr(xs), None) == 1
    a2 = a.copy(); a2[n - 1, 0] ^= 1
    assert L.zkt_bp_inner_product_argument(n, ptr(gg), ptr(hh), ptr(u), ptr(P), ptr(a2), ptr(b), ptr(xs), None) == 0


@pytest.mark.parametrize("n,value", [(4, 9), (64, 0xDEADBEEFCAFEF00D)])
def test_range_proof_vs_oracle(L, n, value):          # bulletproofs.rs:58-147 / test at :248-282 (n = 4, value 9)
    from test_oracle_protocols import range_proof_instance
    V, aL, gamma, g, h, gg, hh, rnd, u, xs = range_proof_instance(n, value, 40 + n)
    for use_ipa in (0, 1):
        gp, op = np.zeros((5, 9), np.uint64), np.zeros((5, 9), np.uint64)
        want = O.zkto_bp_range_proof(n, ptr(V), ptr(aL), ptr(gamma), ptr(g), ptr(h), ptr(gg), ptr(hh), use_ipa, ptr(rnd), ptr(u), ptr(xs), ptr(op))
        got = L.zkt_bp_range_proof(n, ptr(V), ptr(aL), ptr(gamma), ptr(g), ptr(h), ptr(gg), ptr(hh), use_ipa, ptr(rnd), ptr(u), ptr(xs), ptr(gp))
        assert want == 1 and got == 1
        assert (gp == op).all()                       # A, S, T1, T2, P identical
        bad = aL.copy(); bad[1, 0] ^= 1
        assert L.zkt_bp_range_proof(n, ptr(V), ptr(bad), ptr(gamma), ptr(g), ptr(h), ptr(gg), ptr(hh), use_ipa, ptr(rnd), ptr(u), ptr(xs), None) == 0
    # the same proofs over one context with resident generators: identical points and verdicts, the context reused across calls
    ctx = ctypes.c_void_p(); zk.check(L.zkt_bp_ipa_ctx_create(n, ptr(gg), ptr(hh), ptr(u), ctypes.byref(ctx)))
    for use_ipa in (0, 1, 0):
        gp, op = np.zeros((5, 9), np.uint64), np.zeros((5, 9), np.uint64)
        assert O.zkto_bp_range_proof(n, ptr(V), ptr(aL), ptr(gamma), ptr(g), ptr(h), ptr(gg), ptr(hh), use_ipa, ptr(rnd), ptr(u), ptr(xs), ptr(op)) == 1
        assert L.zkt_bp_range_proof_ctx(ctx, ptr(V), ptr(aL), ptr(gamma), ptr(g), ptr(h), use_ipa, ptr(rnd), ptr(xs), ptr(gp)) == 1
        assert (gp == op).all()
        assert L.zkt_bp_range_proof_ctx(ctx, ptr(V), ptr(bad), ptr(gamma), ptr(g), ptr(h), use_ipa, ptr(rnd), ptr(xs), None) == 0
    L.zkt_bp_ipa_ctx_free(ctx)


def test_groth16_verify_batch_mixed(L):              # f-2: fused 3-pairing verification, many proofs per launch
    A_, B_, C_, wit, l = example_cubic()
    n, m = len(A_), len(wit) - 1
    ui, vi, wi, h, _ = qap_from_r1cs(A_, B_, C_, wit)
    U, V, W = dense(ui, n), dense(vi, n), dense(wi, n)
    rng = SplitMix64(71); trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
    crs, buf = alloc_crs(n, l, m)
    zk.check(L.zkt_groth16_setup(ctypes.byref(crs), ptr(U), ptr(V), ptr(W), *[ptr(t) for t in trap]))
    wires = ints_to_arr(wit, 4); H = ints_to_arr(h, 4)
    k = 6
    As, Bs, Cs = np.zeros((k, G1W), np.uint64), np.zeros((k, G2W), np.uint64), np.zeros((k, G1W), np.uint64)
    for i in range(k):                                  # k proofs of the same statement with different prover randomness
        r, s = fr(rng.below(R - 1) + 1), fr(rng.below(R - 1) + 1)
        zk.check(L.zkt_groth16_prove(ctypes.byref(crs), ptr(U), ptr(V), ptr(wires), ptr(H), len(h), ptr(r), ptr(s), ptr(As[i:i + 1]), ptr(Bs[i:i + 1]), ptr(Cs[i:i + 1])))
    stmts = np.repeat(ints_to_arr(wit[:l + 1], 4).reshape(1, -1), k, axis=0).copy()
    stmts[2].reshape(l + 1, 4)[l, 0] ^= 1              # proof 2: wrong statement
    Cs[4, :12] = As[4, :12]                            # proof 4: tampered C
    ok = np.zeros(k, np.uint32)
    zk.check(L.zkt_groth16_verify_batch(ctypes.byref(crs), ptr(As), ptr(Bs), ptr(Cs), ptr(stmts), l + 1, k, ok.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
    assert ok.tolist() == [1, 1, 0, 1, 0, 1]
    for i in range(k):                                  # the oracle's verifier (three separate tate calls) agrees
        assert O.zkto_groth16_verify(ctypes.byref(crs), ptr(As[i:i + 1].copy()), ptr(Bs[i:i + 1].copy()), ptr(Cs[i:i + 1].copy()), ptr(stmts[i:i + 1].copy()), l + 1) == int(ok[i])
    # The large-batch kernel decides on the 63-step loop and compares with the ate counterpart of alpha_beta, which it may do only for a key whose alpha_beta IS
    # tate(alpha, beta): the reference compares against the stored GTPoint (verifier.rs:48).  A key carrying another GT element keeps the value-comparing kernels.
    keep = buf["gt_alpha_beta"].copy()
    g1g = np.zeros((1, G1W), np.uint64); O.zkto_g1_generator(ptr(g1g)); g2g = np.zeros((1, G2W), np.uint64); O.zkto_g2_generator(ptr(g2g))
    zk.check(L.zkt_tate_batch(ptr(g1g), ptr(g2g), ptr(buf["gt_alpha_beta"]), 1))
    zk.check(L.zkt_groth16_verify_batch(ctypes.byref(crs), ptr(As), ptr(Bs), ptr(Cs), ptr(stmts), l + 1, k, ok.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
    want = [O.zkto_groth16_verify(ctypes.byref(crs), ptr(As[i:i + 1].copy()), ptr(Bs[i:i + 1].copy()), ptr(Cs[i:i + 1].copy()), ptr(stmts[i:i + 1].copy()), l + 1) for i in range(k)]
    assert ok.tolist() == want == [0] * k
    buf["gt_alpha_beta"][:] = keep
    # A (then C) on the curve outside G1: the reference evaluates tate(A, B) and the right side and compares, or panics inside tate (verifier.rs:36-53,
    # rational_function.rs:36) — the engine must give that accept / reject / panic, element by element; the oracle's three-tate verifier is the checker.
    for which, arr in (("A", As), ("C", Cs)):
        for label, pt in degenerate_g1_points():
            keep5 = arr[5].copy(); arr[5] = g1_arr([pt])[0]
            want5 = O.zkto_groth16_verify(ctypes.byref(crs), ptr(As[5:6].copy()), ptr(Bs[5:6].copy()), ptr(Cs[5:6].copy()), ptr(stmts[5:6].copy()), l + 1)
            rc = L.zkt_groth16_verify_batch(ctypes.byref(crs), ptr(As), ptr(Bs), ptr(Cs), ptr(stmts), l + 1, k, ok.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
            idx = L.zkt_last_error_index()
            one = L.zkt_groth16_verify(ctypes.byref(crs), ptr(As[5:6].copy()), ptr(Bs[5:6].copy()), ptr(Cs[5:6].copy()), ptr(stmts[5:6].copy()), l + 1)
            if want5 < 0:
                assert rc == ZKT_ERR_INFINITY and idx == 5 and one == -ZKT_ERR_INFINITY, (which, label)
            else:
                assert rc == ZKT_OK and ok.tolist() == [1, 1, 0, 1, 0, want5] and one == want5, (which, label)
            arr[5] = keep5
    L.zkt_verify_set_fail_closed(1)                  # the explicit switch back to round 3's behaviour: such a proof is rejected without the reference's evaluation
    try:
        A5 = As[5].copy(); As[5] = g1_arr([degenerate_g1_points()[0][1]])[0]
        zk.check(L.zkt_groth16_verify_batch(ctypes.byref(crs), ptr(As), ptr(Bs), ptr(Cs), ptr(stmts), l + 1, k, ok.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
        assert ok.tolist() == [1, 1, 0, 1, 0, 0]
        As[5] = A5
    finally:
        L.zkt_verify_set_fail_closed(0)
    # G2 arguments outside the subgroup: the fast kernels need B, gamma, delta in G2 (127-step loop) and redo such proofs on the 255-step kernels —
    # a proof whose B is a twist point outside G2, then a verifying key whose gamma is one: verdicts as the oracle's three-tate verifier gives them
    rng2 = SplitMix64(72)
    Bs[1] = g2_arr([to_abi_g2(py_twist_point(rng2))])[0]
    for swap_gamma in (False, True):
        if swap_gamma: buf["g2_gamma"][0] = g2_arr([to_abi_g2(py_twist_point(rng2))])[0]
        zk.check(L.zkt_groth16_verify_batch(ctypes.byref(crs), ptr(As), ptr(Bs), ptr(Cs), ptr(stmts), l + 1, k, ok.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
        want = [O.zkto_groth16_verify(ctypes.byref(crs), ptr(As[i:i + 1].copy()), ptr(Bs[i:i + 1].copy()), ptr(Cs[i:i + 1].copy()), ptr(stmts[i:i + 1].copy()), l + 1) for i in range(k)]
        assert ok.tolist() == want and ok[1] == 0


@pytest.mark.parametrize("name,W,order,gen_fn,n", [("g2", G2W, R, "zkto_g2_generator", 1 << 13), ("secp", 9, SECP_N, "zkto_secp_generator", 1 << 14)])
def test_g2_secp_msm_large_by_linearity(L, name, W, order, gen_fn, n):
    """Pippenger over Fq2 / secp256k1 at a size the oracle cannot sum term by term: bases k_i*G, so the MSM must equal
    (sum k_i s_i mod order)*G — python integers plus one oracle scalar multiplication."""
    import time
    rng = np.random.Generator(np.random.PCG64(123))
    ks = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64); ks[:, 3] >>= np.uint64(2)
    ss = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64); ss[:, 3] >>= np.uint64(2)
    ss[5] = 0; ss[6] = [1, 0, 0, 0]
    g = np.zeros((1, W), np.uint64); getattr(O, gen_fn)(ptr(g))
    bases = np.zeros((n, W), np.uint64)
    zk.check(getattr(L, f"zkt_{name}_mul_batch")(ptr(np.repeat(g, n, axis=0)), ptr(ks), 4, ptr(bases), n))
    got = np.zeros((1, W), np.uint64)
    t0 = time.perf_counter()
    zk.check(getattr(L, f"zkt_{name}_msm")(ptr(bases), ptr(ss), n, ptr(got)))
    dt = time.perf_counter() - t0
    tot = sum(limbs_to_int(a) * limbs_to_int(b) for a, b in zip(ks, ss)) % order
    want = np.zeros((1, W), np.uint64)
    assert getattr(O, f"zkto_{name}_mul_batch")(ptr(g), ptr(ints_to_arr([tot], 4)), 4, ptr(want), 1, 1) == 0
    assert (got == want).all()
    print(f"{name} one-shot MSM n={n}: {dt*1e3:.1f} ms (incl. upload + window-multiple table build)")


@pytest.mark.parametrize("name,W,order,gen_fn,pw", [("g1", G1W, R, "zkto_g1_generator", zk.G1_PARTIAL_WORDS), ("g2", G2W, R, "zkto_g2_generator", zk.G2_PARTIAL_WORDS),
                                                    ("secp", 9, SECP_N, "zkto_secp_generator", zk.SECP_PARTIAL_WORDS)])
def test_sharded_msm_partials_combine(L, name, W, order, gen_fn, pw):
    """The multi-GPU combine step on one card: split an MSM into index-range shards with resident bases, take each
    shard's opaque Jacobian partial, sum the partials with zkt_*_jac_sum_dev -> identical to the unsharded MSM (and the oracle)."""
    import torch
    n, shards = 3000, 3
    rng = np.random.Generator(np.random.PCG64(77))
    ks = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64); ks[:, 3] >>= np.uint64(2)
    ss = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64); ss[:, 3] >>= np.uint64(2)
    g = np.zeros((1, W), np.uint64); getattr(O, gen_fn)(ptr(g))
    bases = np.zeros((n, W), np.uint64)
    zk.check(getattr(L, f"zkt_{name}_mul_batch")(ptr(np.repeat(g, n, axis=0)), ptr(ks), 4, ptr(bases), n))
    whole = np.zeros((1, W), np.uint64); zk.check(getattr(L, f"zkt_{name}_msm")(ptr(bases), ptr(ss), n, ptr(whole)))
    parts = torch.zeros((shards, pw), dtype=torch.int32, device="cuda")
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    from sharded import shard_range
    for k in range(shards):
        lo, hi = shard_range(n, k, shards)
        h = ctypes.c_void_p(); zk.check(getattr(L, f"zkt_{name}_bases_upload")(ptr(bases[lo:hi].copy()), hi - lo, ctypes.byref(h)))
        d_s = torch.from_numpy(ss[lo:hi].copy().view(np.int64)).cuda()
        out = np.zeros((1, W), np.uint64)
        zk.check(getattr(L, f"zkt_{name}_msm_dev")(h, vp(d_s), hi - lo, None, ptr(out), vp(parts[k])))
        getattr(L, f"zkt_{name}_bases_free")(h)
    torch.cuda.synchronize()
    got = np.zeros((1, W), np.uint64)
    zk.check(getattr(L, f"zkt_{name}_jac_sum_dev")(vp(parts), shards, None, ptr(got)))
    assert (got == whole).all()
    tot = sum(limbs_to_int(a) * limbs_to_int(b) for a, b in zip(ks, ss)) % order
    want = np.zeros((1, W), np.uint64)
    assert getattr(O, f"zkto_{name}_mul_batch")(ptr(g), ptr(ints_to_arr([tot], 4)), 4, ptr(want), 1, 1) == 0
    assert (got == want).all()


@pytest.mark.parametrize("name,W", [("g1", G1W), ("g2", G2W), ("secp", 9)])
def test_msm_eight_slots_in_flight(L, name, W):
    """zkt_*_msm_submit / _collect with every pipeline slot busy on one small resident base set (below 2^19 terms each slot runs its whole
    MSM on its own stream): eight different scalar vectors, some with zero halves like the inner-product argument's, collected in a shuffled
    order, each identical to the one-shot MSM of the same inputs (itself checked against the oracle by the tests above)."""
    import torch
    n, slots = 700, 8
    rng = np.random.Generator(np.random.PCG64(4242))
    order = SECP_N if name == "secp" else R
    g = np.zeros((1, W), np.uint64); getattr(O, f"zkto_{name}_generator")(ptr(g))
    ks = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64); ks[:, 3] >>= np.uint64(2)
    bases = np.zeros((n, W), np.uint64)
    zk.check(getattr(L, f"zkt_{name}_mul_batch")(ptr(np.repeat(g, n, axis=0)), ptr(ks), 4, ptr(bases), n))
    bases[5] = 0; bases[5, W - 1] = 1                          # a base at infinity
    h = ctypes.c_void_p(); zk.check(getattr(L, f"zkt_{name}_bases_upload")(ptr(bases), n, ctypes.byref(h)))
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    try:
        for rnd in range(2):                                    # slots are reusable
            ss = [rand_scalars(9000 + 10 * rnd + k, n, order) for k in range(slots)]
            ss[1][: n // 2] = 0; ss[2][n // 2:] = 0; ss[3][:] = 0; ss[4][:, 1:] = 0      # zero halves, all zero, 64-bit scalars
            d = [torch.from_numpy(x.view(np.int64)).cuda() for x in ss]
            for k in range(slots): zk.check(getattr(L, f"zkt_{name}_msm_submit")(h, vp(d[k]), n, None, k))
            assert getattr(L, f"zkt_{name}_msm_submit")(h, vp(d[0]), n, None, 0) != 0           # a busy slot is refused
            for k in [3, 0, 7, 1, 6, 2, 5, 4]:
                got, want = np.zeros((1, W), np.uint64), np.zeros((1, W), np.uint64)
                zk.check(getattr(L, f"zkt_{name}_msm_collect")(h, k, ptr(got), None))
                zk.check(getattr(L, f"zkt_{name}_msm")(ptr(bases), ptr(ss[k]), n, ptr(want)))
                assert (got == want).all(), (name, rnd, k)
    finally:
        getattr(L, f"zkt_{name}_bases_free")(h)


@pytest.mark.parametrize("name,W,order,gen_fn", [("g1", G1W, R, "zkto_g1_generator"), ("g2", G2W, R, "zkto_g2_generator"), ("secp", 9, SECP_N, "zkto_secp_generator")])
def test_msm_adversarial_pool(L, name, W, order, gen_fn):
    """Bases drawn from a tiny pool {P1..P3, -P1, -P2, infinity} with scalars from {0, 1, 2, order-1, a few random values}: almost every
    bucket then sees equal points (the doubling branch of the mixed add, macros.rs:57-108), opposite points (P + (-P) = infinity,
    macros.rs:53-56) and infinity operands, in the accumulate, merge and reduce kernels alike.  Checked against the oracle's sequential sum."""
    n = 600
    rng = SplitMix64(4100)
    g = np.zeros((1, W), np.uint64); getattr(O, gen_fn)(ptr(g))
    pool = np.zeros((6, W), np.uint64)
    assert getattr(O, f"zkto_{name}_mul_batch")(ptr(np.repeat(g, 3, axis=0)), ptr(ints_to_arr([rng.below(order - 1) + 1 for _ in range(3)], 4)), 4, ptr(pool[:3]), 3, 3) == 0
    assert getattr(O, f"zkto_{name}_mul_batch")(ptr(pool[:2].copy()), ptr(ints_to_arr([order - 1] * 2, 4)), 4, ptr(pool[3:5]), 2, 2) == 0     # -P = (order-1) P
    pool[5] = 0; pool[5, W - 1] = 1
    few = [rng.below(order) for _ in range(3)]
    choices = [0, 1, 2, order - 1, 1, 2] + few
    bases = np.stack([pool[rng.below(6)] for _ in range(n)])
    ss = [choices[rng.below(len(choices))] for _ in range(n)]
    sc = ints_to_arr(ss, 4)
    got = np.zeros((1, W), np.uint64); zk.check(getattr(L, f"zkt_{name}_msm")(ptr(bases), ptr(sc), n, ptr(got)))
    tmp = np.zeros_like(bases); assert getattr(O, f"zkto_{name}_mul_batch")(ptr(bases), ptr(sc), 4, ptr(tmp), n, 8) == 0
    acc = np.zeros((1, W), np.uint64); acc[0, W - 1] = 1
    for i in range(n):
        nxt = np.zeros_like(acc); assert getattr(O, f"zkto_{name}_add_batch")(ptr(acc), ptr(tmp[i:i + 1].copy()), ptr(nxt), 1) == 0; acc = nxt
    assert (got == acc).all()


@pytest.mark.parametrize("n", [4097, 70001, 300000])
def test_g1_one_shot_msm_sizes_by_linearity(L, n):
    """One-shot calls use the table-free form whose window size follows n (msm_plan_direct: c = 9..16): odd sizes on both sides of the
    window-size steps, checked by linearity — bases k_i*G, so the sum is (sum k_i s_i mod r)*G in python integers."""
    rng = np.random.Generator(np.random.PCG64(2024 + n))
    ks = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64); ks[:, 3] >>= np.uint64(2)
    ss = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64); ss[:, 3] >>= np.uint64(2)
    ss[0] = 0; ss[1] = [1, 0, 0, 0]; ss[n - 1] = int_to_limbs(R - 1, 4)
    g = g1_arr([G1_GEN])
    bases = np.zeros((n, G1W), np.uint64)
    zk.check(L.zkt_g1_mul_batch(ptr(np.repeat(g, n, axis=0)), ptr(ks), 4, ptr(bases), n))
    got = np.zeros((1, G1W), np.uint64)
    zk.check(L.zkt_g1_msm(ptr(bases), ptr(ss), n, ptr(got)))
    tot = sum(limbs_to_int(a) * limbs_to_int(b) for a, b in zip(ks, ss)) % R
    assert (got == g1_arr([py_g1_mul(G1_GEN, tot)])).all()


def test_groth16_vk_prepare(L):
    """zkt_groth16_vk_prepare builds a key's entry for the 63-step loop ahead of the first verification (verifier.rs:30-54 builds nothing per key; crs.rs:137-139 computes
    alpha_beta once).  Decisions do not depend on it: a prepared key accepts its proof and rejects a wrong statement from the first call on; a key whose stored alpha_beta is NOT
    tate(alpha, beta) is prepared without error, never qualifies for the 63-step loop, and is decided against the stored element exactly as the oracle decides; shapes are checked."""
    A_, B_, C_, wit, l = example_cubic()
    n, m = len(A_), len(wit) - 1
    ui, vi, wi, h, _ = qap_from_r1cs(A_, B_, C_, wit)
    U, V, W = dense(ui, n), dense(vi, n), dense(wi, n)
    wires, H = ints_to_arr(wit, 4), ints_to_arr(h, 4)
    stmt = ints_to_arr(wit[:l + 1], 4)
    bad = stmt.copy(); bad[l, 0] ^= np.uint64(1)
    rng = SplitMix64(8800); trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
    crs, buf = alloc_crs(n, l, m)
    zk.check(L.zkt_groth16_setup(ctypes.byref(crs), ptr(U), ptr(V), ptr(W), *[ptr(t) for t in trap]))
    pf = (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
    zk.check(L.zkt_groth16_prove(ctypes.byref(crs), ptr(U), ptr(V), ptr(wires), ptr(H), len(h), ptr(fr(rng.below(R - 1) + 1)), ptr(fr(rng.below(R - 1) + 1)), *[ptr(x) for x in pf]))
    L.zkt_groth16_vk_prepare.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert L.zkt_groth16_vk_prepare(ctypes.byref(crs), l + 2) == ZKT_ERR_SHAPE               # more statement wires than the key has
    assert L.zkt_groth16_vk_prepare(None, l + 1) == ZKT_ERR_SHAPE
    assert L.zkt_groth16_vk_prepare(ctypes.byref(crs), 0) == 0                               # nothing to prepare
    assert L.zkt_groth16_vk_prepare(ctypes.byref(crs), l + 1) == 0
    assert L.zkt_groth16_vk_prepare(ctypes.byref(crs), l + 1) == 0                           # again: the entry exists
    for st, want in ((stmt, 1), (bad, 0), (stmt, 1)):
        assert L.zkt_groth16_verify(ctypes.byref(crs), *[ptr(x) for x in pf], ptr(st), l + 1) == want
        assert O.zkto_groth16_verify(ctypes.byref(crs), *[ptr(x) for x in pf], ptr(st), l + 1) == want
    # the same key with another GT element in place of alpha_beta (here: its square): preparing succeeds, the verifier compares against THAT element like the reference does
    crs2, buf2 = alloc_crs(n, l, m)
    zk.check(L.zkt_groth16_setup(ctypes.byref(crs2), ptr(U), ptr(V), ptr(W), *[ptr(t) for t in trap]))
    gt = buf2["gt_alpha_beta"]
    sq = np.zeros_like(gt); zk.check(L.zkt_fq12_mul_batch(ptr(gt.copy()), ptr(gt.copy()), ptr(sq), 1)); gt[:] = sq
    assert L.zkt_groth16_vk_prepare(ctypes.byref(crs2), l + 1) == 0
    for st in (stmt, bad):
        want = O.zkto_groth16_verify(ctypes.byref(crs2), *[ptr(x) for x in pf], ptr(st), l + 1)
        assert want == 0                                                                      # the proof was made for alpha_beta, not for its square
        for _ in range(2): assert L.zkt_groth16_verify(ctypes.byref(crs2), *[ptr(x) for x in pf], ptr(st), l + 1) == want


def test_groth16_verify_from_four_threads(L):
    """Verifier::verify (verifier.rs:30-54) from four caller threads at once, three keys between them, none of them seen before: the key cache is shared state (an entry is
    begun by the first caller that shows a key — guard stream, side stream, pinned verdict — while another thread may look the same key up, find it busy and take the 127-step
    kernels).  Every decision must be the single-threaded one: accept for the key's own proof, reject for a wrong statement and for another key's proof."""
    import threading
    A_, B_, C_, wit, l = example_cubic()
    n, m = len(A_), len(wit) - 1
    ui, vi, wi, h, _ = qap_from_r1cs(A_, B_, C_, wit)
    U, V, W = dense(ui, n), dense(vi, n), dense(wi, n)
    wires, H = ints_to_arr(wit, 4), ints_to_arr(h, 4)
    stmt = ints_to_arr(wit[:l + 1], 4)
    bad = stmt.copy(); bad[l, 0] ^= np.uint64(1)
    keys = []
    for k in range(3):
        rng = SplitMix64(7700 + k); trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
        crs, buf = alloc_crs(n, l, m)
        zk.check(L.zkt_groth16_setup(ctypes.byref(crs), ptr(U), ptr(V), ptr(W), *[ptr(t) for t in trap]))
        pf = (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
        zk.check(L.zkt_groth16_prove(ctypes.byref(crs), ptr(U), ptr(V), ptr(wires), ptr(H), len(h), ptr(fr(rng.below(R - 1) + 1)), ptr(fr(rng.below(R - 1) + 1)), *[ptr(x) for x in pf]))
        keys.append((crs, buf, pf))
    errors = []
    start = threading.Barrier(4)
    def worker(t):
        try:
            start.wait()
            for it in range(6):
                k = (t + it) % 3
                crs, pf, other = keys[k][0], keys[k][2], keys[(k + 1) % 3][2]
                st, bd = stmt.copy(), bad.copy()
                got = (L.zkt_groth16_verify(ctypes.byref(crs), *[ptr(x) for x in pf], ptr(st), l + 1),
                       L.zkt_groth16_verify(ctypes.byref(crs), *[ptr(x) for x in pf], ptr(bd), l + 1),
                       L.zkt_groth16_verify(ctypes.byref(crs), *[ptr(x) for x in other], ptr(st), l + 1))
                if got != (1, 0, 0): errors.append((t, it, k, got))
        except Exception as e:                                              # a thread's exception must fail the test, not vanish
            errors.append((t, repr(e)))
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errors, errors
    # a batch of 4,096 proofs on one of those keys right afterwards (the per-lane kernel reads the same entry)
    k = 4096
    pf = keys[1][2]
    As, Bs, Cs = np.repeat(pf[0], k, axis=0), np.repeat(pf[1], k, axis=0), np.repeat(pf[2], k, axis=0)
    sts = np.repeat(stmt.reshape(1, -1), k, axis=0).copy(); sts[17] = bad.reshape(-1)
    ok = np.zeros(k, np.uint32)
    zk.check(L.zkt_groth16_verify_batch(ctypes.byref(keys[1][0]), ptr(As), ptr(Bs), ptr(Cs), ptr(sts), l + 1, k, ok.ctypes.data))
    want = np.ones(k, np.uint32); want[17] = 0
    assert (ok == want).all()


def test_groth16_verify_with_several_keys(L):
    """The verifier keeps what a key contributes to the 63-step loop (line tables of gamma and delta, the ate counterpart of alpha_beta, statement tables) for the last four
    keys, builds an entry at the FIRST sight of a key, beside the call that brought it (that call is served by the 127-step kernels), and evicts the least recently used one.  Five keys in rotation, each
    verified three times in a row and revisited after its entry has been evicted: every decision equals the oracle's, for the valid proof and for a wrong statement."""
    A_, B_, C_, wit, l = example_cubic()
    n, m = len(A_), len(wit) - 1
    ui, vi, wi, h, _ = qap_from_r1cs(A_, B_, C_, wit)
    U, V, W = dense(ui, n), dense(vi, n), dense(wi, n)
    wires, H = ints_to_arr(wit, 4), ints_to_arr(h, 4)
    stmt = ints_to_arr(wit[:l + 1], 4)
    bad = stmt.copy(); bad[l, 0] ^= np.uint64(1)
    keys = []
    for k in range(5):
        rng = SplitMix64(900 + k); trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
        crs, buf = alloc_crs(n, l, m)
        zk.check(L.zkt_groth16_setup(ctypes.byref(crs), ptr(U), ptr(V), ptr(W), *[ptr(t) for t in trap]))
        pf = (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
        zk.check(L.zkt_groth16_prove(ctypes.byref(crs), ptr(U), ptr(V), ptr(wires), ptr(H), len(h), ptr(fr(rng.below(R - 1) + 1)), ptr(fr(rng.below(R - 1) + 1)), *[ptr(x) for x in pf]))
        keys.append((crs, buf, pf))
    seen = {}
    def both(k, j, st, tag):                                              # key k, the proof made for key j; the oracle's decision is computed once per case (three tate() calls)
        crs, pf = keys[k][0], keys[j][2]
        if (k, j, tag) not in seen: seen[(k, j, tag)] = O.zkto_groth16_verify(ctypes.byref(crs), *[ptr(x) for x in pf], ptr(st), l + 1)
        return (seen[(k, j, tag)], L.zkt_groth16_verify(ctypes.byref(crs), *[ptr(x) for x in pf], ptr(st), l + 1))
    for k in (0, 1, 2, 3, 4, 0, 2, 4, 1):
        for _ in range(3):
            assert both(k, k, stmt, "ok") == (1, 1)
        assert both(k, k, bad, "bad") == (0, 0)
        assert both(k, (k + 1) % 5, stmt, "other") == (0, 0)              # a proof made for another key
