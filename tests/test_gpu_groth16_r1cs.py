"""the scalable Groth16 path (sparse R1CS on the reference's domain {1..n}, no coefficient-form
polynomials) must produce the verifying key and the proof points of the reference algorithm bit for bit.  The checker is the
oracle's restatement of CRS::new / Prover::prove (crs.rs:49-146, prover.rs:96-147) on the dense QAP of tests/qap_util.py."""
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


def _r1cs_setup(L, mats, n, l, m, trap):
    structs = [sparse_struct(*sparse_rows(M)) if isinstance(M, list) else sparse_struct(*M) for M in mats]
    vk, vbuf = alloc_crs(1, l, m)                      # xi / xt_by_delta arrays are not produced by this path
    pk = ctypes.c_void_p()
    zk.check(L.zkt_groth16_setup_r1cs(n, l, m, *[ctypes.addressof(s) for s in structs], *[t.ctypes.data for t in trap], ctypes.addressof(vk), ctypes.addressof(pk)))
    return vk, vbuf, pk


@pytest.mark.parametrize("case", ["cubic", "chain1", "chain2", "chain3", "chain16", "chain61", "bits61"])
def test_r1cs_path_matches_reference_algorithm(L, case):
    A, B, C, wit, l = example_cubic() if case == "cubic" else bits_circuit(int(case[4:])) if case.startswith("bits") else chain_circuit(int(case[5:]))
    n, m = len(A), len(wit) - 1
    ui, vi, wi, h, _ = qap_from_r1cs(A, B, C, wit)
    U, V, W = dense(ui, n), dense(vi, n), dense(wi, n)
    rng = SplitMix64(4242 + n)
    trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
    r, s = fr(rng.below(R - 1) + 1), fr(rng.below(R - 1) + 1)
    wires, H = ints_to_arr(wit, 4), ints_to_arr(h, 4)
    ocrs, obuf = alloc_crs(n, l, m)
    assert O.zkto_groth16_setup(ctypes.byref(ocrs), ptr(U), ptr(V), ptr(W), *[ptr(t) for t in trap]) == 0
    op = (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
    assert O.zkto_groth16_prove(ctypes.byref(ocrs), ptr(U), ptr(V), ptr(wires), ptr(H), len(h), ptr(r), ptr(s), 1 if n <= 5 else 0, *[ptr(x) for x in op]) == 0
    vk, vbuf, pk = _r1cs_setup(L, (A, B, C), n, l, m, trap)
    for k in ("g1_alpha", "g1_beta", "g1_delta", "g1_uvw_stmt", "g2_beta", "g2_gamma", "g2_delta", "gt_alpha_beta"):
        assert (obuf[k] == vbuf[k]).all(), f"verifying-key field {k} differs"
    if m > l: assert (obuf["g1_uvw_wit"] == vbuf["g1_uvw_wit"]).all()
    gp = (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
    for _ in range(2):                                 # the proving key is reusable
        zk.check(L.zkt_groth16_prove_r1cs(pk, wires.ctypes.data, r.ctypes.data, s.ctypes.data, *[x.ctypes.data for x in gp]))
        for a, b, name in zip(op, gp, "ABC"):
            assert (a == b).all(), f"proof element {name} differs"
    stmt = ints_to_arr(wit[:l + 1], 4)
    assert L.zkt_groth16_verify(ctypes.byref(vk), ptr(gp[0]), ptr(gp[1]), ptr(gp[2]), ptr(stmt), l + 1) == 1
    L.zkt_groth16_pk_free(pk)


def test_one_constraint_key_proves_again_after_device_memory_was_freed(L):
    """A one-constraint key has an EMPTY quotient base set (n - 1 = 0 terms): its zero-term MSM is captured as a graph like any other, and that graph must hold kernel
    nodes only — a runtime-owned memset node faults on the first replay after any later hipFree (graph replay).  The counters of a zero-term sort are cleared by
    k_zero_words whatever their alignment (zkt_msm.hip zero_async; round-3 advisor finding): prove, free device memory elsewhere in the process, prove again — same proof."""
    import torch
    A, B, C, wit, l = chain_circuit(1)
    n, m = len(A), len(wit) - 1
    rng = SplitMix64(4243)
    trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
    r, s = fr(rng.below(R - 1) + 1), fr(rng.below(R - 1) + 1)
    wires = ints_to_arr(wit, 4)
    vk, vbuf, pk = _r1cs_setup(L, (A, B, C), n, l, m, trap)
    proofs = []
    for rep in range(4):
        gp = (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
        zk.check(L.zkt_groth16_prove_r1cs(pk, wires.ctypes.data, r.ctypes.data, s.ctypes.data, *[x.ctypes.data for x in gp]))
        proofs.append(gp)
        junk = torch.empty(1 << 24, dtype=torch.int64, device="cuda"); junk.fill_(rep); torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()      # a hipFree between replays
    for gp in proofs[1:]:
        assert all((a == b).all() for a, b in zip(proofs[0], gp))
    assert L.zkt_groth16_verify(ctypes.byref(vk), ptr(proofs[0][0]), ptr(proofs[0][1]), ptr(proofs[0][2]), ptr(ints_to_arr(wit[:l + 1], 4)), l + 1) == 1
    L.zkt_groth16_pk_free(pk)


@pytest.mark.parametrize("n", [1000, 5000])
def test_r1cs_path_larger_sizes_verify(L, n):
    """beyond the sizes the quadratic oracle can follow: the proof must verify (verifier.rs:30-54), and must stop
    verifying when the statement or the witness is wrong."""
    mats, wires, l, m = chain_circuit_sparse(n, seed=11)
    rng = SplitMix64(777 + n)
    trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
    r, s = fr(rng.below(R - 1) + 1), fr(rng.below(R - 1) + 1)
    vk, vbuf, pk = _r1cs_setup(L, mats, n, l, m, trap)
    gp = (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
    zk.check(L.zkt_groth16_prove_r1cs(pk, wires.ctypes.data, r.ctypes.data, s.ctypes.data, *[x.ctypes.data for x in gp]))
    stmt = wires[:l + 1].copy()
    assert L.zkt_groth16_verify(ctypes.byref(vk), ptr(gp[0]), ptr(gp[1]), ptr(gp[2]), ptr(stmt), l + 1) == 1
    bad = stmt.copy(); bad[1, 0] ^= np.uint64(1)
    assert L.zkt_groth16_verify(ctypes.byref(vk), ptr(gp[0]), ptr(gp[1]), ptr(gp[2]), ptr(bad), l + 1) == 0
    w2 = wires.copy(); w2[5, 0] ^= np.uint64(2)       # a witness that no longer satisfies the constraints
    zk.check(L.zkt_groth16_prove_r1cs(pk, w2.ctypes.data, r.ctypes.data, s.ctypes.data, *[x.ctypes.data for x in gp]))
    assert L.zkt_groth16_verify(ctypes.byref(vk), ptr(gp[0]), ptr(gp[1]), ptr(gp[2]), ptr(stmt), l + 1) == 0
    L.zkt_groth16_pk_free(pk)


def _g2_gen_c0c1():
    (x1, x0), (y1, y0) = G2_GEN
    return ((x0, x1), (y0, y1))


@pytest.mark.parametrize("n", [1000, 4097, 1 << 16])
def test_r1cs_proof_points_equal_python_integer_multiples(L, n):
    """Beyond the oracle's reach (its dense prover is O(m n)) the proof POINTS are still checkable: with the trapdoor injected the discrete logarithms of
    A, B, C are O(n) python-integer arithmetic (qap_util.groth16_proof_scalars, checked against the coefficient-form prover on the CPU by
    tests/test_r1cs_domain_math.py), and the points are those multiples of the generators in python integers — no HIP code and no oracle code on the
    checking side (prover.rs:96-147, crs.rs:49-146).  n = 4097: the unsharded quotient needs 4096 values, i.e. TWO input blocks of 4096 (the second holds one value)."""
    mats, wires, l, m = chain_circuit_sparse(n, seed=23)
    rng = SplitMix64(4321 + n)
    trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
    r, s = fr(rng.below(R - 1) + 1), fr(rng.below(R - 1) + 1)
    vk, vbuf, pk = _r1cs_setup(L, mats, n, l, m, trap)
    gp = (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
    zk.check(L.zkt_groth16_prove_r1cs(pk, wires.ctypes.data, r.ctypes.data, s.ctypes.data, *[x.ctypes.data for x in gp]))
    L.zkt_groth16_pk_free(pk)
    As, Bs, Cs = groth16_proof_scalars(mats, wires, l, trap, r, s)
    assert (gp[0] == g1_arr([py_g1_mul(G1_GEN, As)])).all(), "A"
    assert (gp[1] == g2_arr([to_abi_g2(py_g2_mul(_g2_gen_c0c1(), Bs))])).all(), "B"
    assert (gp[2] == g1_arr([py_g1_mul(G1_GEN, Cs)])).all(), "C"


@pytest.mark.parametrize("n,shards", [(200, 3), (3000, 5), (2500, 8), (12, 11), (3, 3)])
def test_r1cs_sharded_proof_equals_unsharded(L, n, shards):
    """BASELINE config 4 on one card: the shards of the resident base sets each produce their three Jacobian partials; summing
    them (the all_gather + combine step of the multi-GPU run) gives the unsharded proof bit for bit.  Every rank evaluates only its own
    range of the quotient's values (blocked convolution, csrc/zkt_groth16_r1cs.hip k_recip_blocks): (200, 3) is one input block per rank,
    (3000, 5) three blocks of 1024 with a ragged last one, (2500, 8) likewise with eight ranks, (12, 11) ranges of one term (one quotient value per rank), (3, 3) a rank with NO quotient value (two values, three ranks)."""
    import torch
    mats, wires, l, m = chain_circuit_sparse(n, seed=5)
    rng = SplitMix64(999)
    trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
    r, s = fr(rng.below(R - 1) + 1), fr(rng.below(R - 1) + 1)
    vk, vbuf, pk = _r1cs_setup(L, mats, n, l, m, trap)
    want = (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
    zk.check(L.zkt_groth16_prove_r1cs(pk, wires.ctypes.data, r.ctypes.data, s.ctypes.data, *[x.ctypes.data for x in want]))
    L.zkt_groth16_pk_free(pk)
    d_w = torch.from_numpy(wires.view(np.int64)).cuda()
    parts = torch.zeros((shards, zk.GROTH16_PARTIAL_WORDS), dtype=torch.int32, device="cuda")
    structs = [sparse_struct(*M) for M in mats]
    for k in range(shards):
        vk2, vbuf2 = alloc_crs(1, l, m); pk2 = ctypes.c_void_p()
        zk.check(L.zkt_groth16_setup_r1cs_sharded(n, l, m, *[ctypes.addressof(x) for x in structs], *[t.ctypes.data for t in trap], k, shards,
                                                   ctypes.addressof(vk2), ctypes.addressof(pk2)))
        zk.check(L.zkt_groth16_prove_r1cs_partials(pk2, d_w.data_ptr(), r.ctypes.data, s.ctypes.data, parts[k].data_ptr()))
        # a shard cannot produce affine points on its own
        assert L.zkt_groth16_prove_r1cs_dev(pk2, d_w.data_ptr(), r.ctypes.data, s.ctypes.data, *[x.ctypes.data for x in want]) == ZKT_ERR_SHAPE
        L.zkt_groth16_pk_free(pk2)
    torch.cuda.synchronize()
    a, b = zk.G1_PARTIAL_WORDS, zk.G1_PARTIAL_WORDS + zk.G2_PARTIAL_WORDS
    got = (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    pa, pb, pc = parts[:, :a].contiguous(), parts[:, a:b].contiguous(), parts[:, b:].contiguous()
    zk.check(L.zkt_g1_jac_sum_dev(vp(pa), shards, None, ptr(got[0])))
    zk.check(L.zkt_g2_jac_sum_dev(vp(pb), shards, None, ptr(got[1])))
    zk.check(L.zkt_g1_jac_sum_dev(vp(pc), shards, None, ptr(got[2])))
    for x, y, name in zip(want, got, "ABC"):
        assert (x == y).all(), f"sharded proof element {name} differs"


def test_r1cs_path_all_wires_public(L):
    """l = m: no witness wires, so the uvw_wit part of the C base set is empty (prover.rs:127-131 loops over nothing)."""
    A, B, C, wit, _ = chain_circuit(3)
    n, m = len(A), len(wit) - 1
    l = m
    ui, vi, wi, h, _ = qap_from_r1cs(A, B, C, wit)
    U, V, W = dense(ui, n), dense(vi, n), dense(wi, n)
    rng = SplitMix64(31337)
    trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
    r, s = fr(rng.below(R - 1) + 1), fr(rng.below(R - 1) + 1)
    wires, H = ints_to_arr(wit, 4), ints_to_arr(h, 4)
    ocrs, obuf = alloc_crs(n, l, m)
    assert O.zkto_groth16_setup(ctypes.byref(ocrs), ptr(U), ptr(V), ptr(W), *[ptr(t) for t in trap]) == 0
    op = (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
    assert O.zkto_groth16_prove(ctypes.byref(ocrs), ptr(U), ptr(V), ptr(wires), ptr(H), len(h), ptr(r), ptr(s), 1, *[ptr(x) for x in op]) == 0
    vk, vbuf, pk = _r1cs_setup(L, (A, B, C), n, l, m, trap)
    assert (obuf["g1_uvw_stmt"] == vbuf["g1_uvw_stmt"]).all()
    gp = (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
    zk.check(L.zkt_groth16_prove_r1cs(pk, wires.ctypes.data, r.ctypes.data, s.ctypes.data, *[x.ctypes.data for x in gp]))
    for a, b, name in zip(op, gp, "ABC"):
        assert (a == b).all(), f"proof element {name} differs"
    assert L.zkt_groth16_verify(ctypes.byref(vk), ptr(gp[0]), ptr(gp[1]), ptr(gp[2]), ptr(wires), l + 1) == 1
    L.zkt_groth16_pk_free(pk)


def test_r1cs_pipelined_proofs_equal_sequential(L):
    """zkt_groth16_prove_r1cs_submit / _collect: two proofs in flight on one key give the very proofs the blocking call gives."""
    import torch
    n = 300
    mats, wires, l, m = chain_circuit_sparse(n, seed=21)
    rng = SplitMix64(555)
    trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
    vk, vbuf, pk = _r1cs_setup(L, mats, n, l, m, trap)
    d_w = torch.from_numpy(wires.view(np.int64)).cuda()
    rs = [(fr(rng.below(R - 1) + 1), fr(rng.below(R - 1) + 1)) for _ in range(5)]
    new = lambda: (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
    seq = []
    for r, s in rs:
        p = new(); zk.check(L.zkt_groth16_prove_r1cs_dev(pk, d_w.data_ptr(), r.ctypes.data, s.ctypes.data, *[x.ctypes.data for x in p])); seq.append(p)
    pip = [new() for _ in rs]
    zk.check(L.zkt_groth16_prove_r1cs_submit(pk, 0, d_w.data_ptr(), rs[0][0].ctypes.data, rs[0][1].ctypes.data))
    assert L.zkt_groth16_prove_r1cs_submit(pk, 0, d_w.data_ptr(), rs[0][0].ctypes.data, rs[0][1].ctypes.data) == ZKT_ERR_SHAPE     # slot still in flight
    for i in range(len(rs)):
        if i + 1 < len(rs):
            zk.check(L.zkt_groth16_prove_r1cs_submit(pk, (i + 1) % 2, d_w.data_ptr(), rs[i + 1][0].ctypes.data, rs[i + 1][1].ctypes.data))
        zk.check(L.zkt_groth16_prove_r1cs_collect(pk, i % 2, *[x.ctypes.data for x in pip[i]]))
    for a, b in zip(seq, pip):
        assert all((x == y).all() for x, y in zip(a, b))
    assert len({seq[i][0].tobytes() for i in range(len(rs))}) == len(rs)          # different r, s -> different proofs
    stmt = wires[:l + 1].copy()
    assert L.zkt_groth16_verify(ctypes.byref(vk), ptr(pip[-1][0]), ptr(pip[-1][1]), ptr(pip[-1][2]), ptr(stmt), l + 1) == 1
    L.zkt_groth16_pk_free(pk)
