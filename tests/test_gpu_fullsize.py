"""The headline code paths at BASELINE.json's sizes, inside the -m gpu suite (one pytest id per BASELINE config):
  config 2  resident-table G1 MSM, n = 2^20 (the c = 20, 13-window plan of `msm_plan`, `k_precompute` table) — linearity against python
            integers for uniform / all-ones / 0-1 scalars, the pipelined submit/collect form and the one-shot host-pointer form;
  config 3  2^16 Tate pairings in one batch — a 64-element sample against the oracle, a whole-batch symmetry property
            e(a G1, b G2) == e(b G1, a G2) over all elements, and e(P,Q) e(-P,Q) == 1 through zkt_pairing_product_check_batch;
  config 4  Groth16 prove + verify at 2^20 constraints (sparse-R1CS path): accept, and two rejects;
  config 5  Bulletproofs range proof over 65,536 bits (64 bits x 1024 values) with and without the inner-product argument: accept / reject.
Size-independent properties replace the oracle where it would take hours."""
import ctypes, importlib, time
import numpy as np
import pytest
from zkt_testlib import *
from qap_util import chain_circuit_sparse, sparse_struct, alloc_crs, groth16_proof_scalars

pytestmark = pytest.mark.gpu
zk = importlib.import_module("zk-toolkit_amd")
O = oracle()


@pytest.fixture(scope="module")
def L():
    zk.init()
    return zk.lib()


def _ints(a):
    """(n, w) u64 -> python ints, fast path"""
    a = np.ascontiguousarray(a)
    return [int.from_bytes(r.tobytes(), "little") for r in a]


def _gen_rows(gen_arr, n):
    return np.repeat(gen_arr, n, axis=0)


def _g1_gen():
    return g1_arr([G1_GEN])


def _g2_gen():
    return g2_arr([G2_GEN])


def test_config2_resident_g1_msm_2p20(L):
    import torch
    n = 1 << 20
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    ks = rand_u64_array(3, (n, 4)); ks[:, 3] >>= np.uint64(2)                    # P_i = k_i G1, k_i < 2^254
    d_k = torch.from_numpy(ks.view(np.int64)).cuda()
    d_gen = torch.from_numpy(_gen_rows(_g1_gen(), n).view(np.int64)).cuda()
    d_bases = torch.empty((n, G1W), dtype=torch.int64, device="cuda")
    zk.check(L.zkt_g1_mul_batch_dev(vp(d_gen), vp(d_k), 4, vp(d_bases), n, None))
    torch.cuda.synchronize(); del d_gen
    h = ctypes.c_void_p()
    zk.check(L.zkt_g1_bases_from_device(vp(d_bases), n, None, ctypes.byref(h)))
    assert L.zkt_g1_bases_len(h) == n
    K = _ints(ks)
    uniform = rand_u64_array(4, (n, 4)); uniform[:, 3] >>= np.uint64(1)          # 255-bit scalars, used as-is (macros.rs:10-21)
    ones = np.zeros((n, 4), np.uint64); ones[:, 0] = 1
    bits = np.zeros((n, 4), np.uint64); bits[:, 0] = rand_u64_array(5, (n,)) & np.uint64(1)
    top = np.zeros((n, 4), np.uint64); top[:] = np.uint64(0xFFFFFFFFFFFFFFFF)    # 2^256 - 1 everywhere: every signed digit carries
    results = {}
    for name, sc in (("uniform", uniform), ("ones", ones), ("bits", bits), ("all-ones-256", top)):
        d_s = torch.from_numpy(sc.view(np.int64)).cuda()
        got = np.zeros((1, G1W), np.uint64)
        zk.check(L.zkt_g1_msm_dev(h, vp(d_s), n, None, ptr(got), None))
        tot = sum(k * s for k, s in zip(K, _ints(sc))) % R
        want = g1_arr([py_g1_mul(G1_GEN, tot)])                                  # python integers only: neither the HIP path nor the oracle
        assert (got == want).all(), f"resident 2^20 MSM, {name} scalars"
        results[name] = got.copy()
    # pipelined form at full size: four MSMs in flight over the same table, alternating scalar sets
    d_u, d_b = torch.from_numpy(uniform.view(np.int64)).cuda(), torch.from_numpy(bits.view(np.int64)).cuda()
    for slot in range(4):
        zk.check(L.zkt_g1_msm_submit(h, vp(d_u if slot % 2 == 0 else d_b), n, None, slot))
    for slot in range(4):
        got = np.zeros((1, G1W), np.uint64)
        zk.check(L.zkt_g1_msm_collect(h, slot, ptr(got), None))
        assert (got == results["uniform" if slot % 2 == 0 else "bits"]).all(), f"pipelined slot {slot}"
    L.zkt_g1_bases_free(h)
    # one-shot host-pointer form (eval_with_g1_hidings called once, polynomial.rs:271-281): the table-free plan at the same size
    bases = d_bases.cpu().numpy().view(np.uint64)
    got = np.zeros((1, G1W), np.uint64)
    zk.check(L.zkt_g1_msm(ptr(bases), ptr(uniform), n, ptr(got)))
    assert (got == results["uniform"]).all(), "one-shot 2^20 MSM"


def test_config4_resident_g2_msm_2p20(L):
    """The B sum of a Groth16 proof at 2^20 constraints is a resident G2 MSM of this size (prover.rs:111,119; eval_with_g2_hidings polynomial.rs:283-293):
    linearity against python integers for uniform / 0-1 / 2^256-1 scalars — the lane-pair accumulate and the Fq2 reduce kernels at the size they run."""
    import torch
    n = 1 << 20
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    ks = rand_u64_array(13, (n, 4)); ks[:, 3] >>= np.uint64(2)
    d_k = torch.from_numpy(ks.view(np.int64)).cuda()
    d_gen = torch.from_numpy(_gen_rows(_g2_gen(), n).view(np.int64)).cuda()
    d_bases = torch.empty((n, G2W), dtype=torch.int64, device="cuda")
    zk.check(L.zkt_g2_mul_batch_dev(vp(d_gen), vp(d_k), 4, vp(d_bases), n, None))
    torch.cuda.synchronize(); del d_gen
    h = ctypes.c_void_p()
    zk.check(L.zkt_g2_bases_from_device(vp(d_bases), n, None, ctypes.byref(h)))
    K = _ints(ks)
    (x1, x0), (y1, y0) = G2_GEN
    gen = ((x0, x1), (y0, y1))
    uniform = rand_u64_array(14, (n, 4)); uniform[:, 3] >>= np.uint64(1)
    bits = np.zeros((n, 4), np.uint64); bits[:, 0] = rand_u64_array(15, (n,)) & np.uint64(1)
    top = np.zeros((n, 4), np.uint64); top[:] = np.uint64(0xFFFFFFFFFFFFFFFF)
    for name, sc in (("uniform", uniform), ("bits", bits), ("all-ones-256", top)):
        d_s = torch.from_numpy(sc.view(np.int64)).cuda()
        got = np.zeros((1, G2W), np.uint64)
        zk.check(L.zkt_g2_msm_dev(h, vp(d_s), n, None, ptr(got), None))
        tot = sum(k * s for k, s in zip(K, _ints(sc))) % R
        want = g2_arr([to_abi_g2(py_g2_mul(gen, tot))])
        assert (got == want).all(), f"resident 2^20 G2 MSM, {name} scalars"
    L.zkt_g2_bases_free(h)


def test_config5_resident_secp_msm_2p17_plus_1(L):
    """Every generator sum of the 65,536-bit range proof is an MSM over [gg | hh | u]: 2^17 + 1 secp256k1 points (bulletproofs.rs:58-147,
    affine_points.rs:25-31,123-144).  Linearity against python integers at exactly that size, uniform / 0-1 / 2^256-1 scalars, with the
    point at infinity and a repeated point among the bases."""
    import torch
    n = (1 << 17) + 1
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    ks = rand_scalars(23, n, SECP_N)
    ks[5] = 0; ks[7] = ks[6]                                                      # an infinity and a repeat among the generators
    gen = secp_arr([SECP_GEN])
    bases = np.zeros((n, 9), np.uint64)
    zk.check(L.zkt_secp_mul_batch(ptr(_gen_rows(gen, n)), ptr(ks), 4, ptr(bases), n))
    assert bases[5, 8] == 1 and (bases[6] == bases[7]).all()
    h = ctypes.c_void_p()
    zk.check(L.zkt_secp_bases_upload(ptr(bases), n, ctypes.byref(h)))
    K = _ints(ks)
    uniform = rand_u64_array(24, (n, 4))                                          # 256-bit scalars, used as-is: some exceed the group order
    bits = np.zeros((n, 4), np.uint64); bits[:, 0] = rand_u64_array(25, (n,)) & np.uint64(1)
    top = np.zeros((n, 4), np.uint64); top[:] = np.uint64(0xFFFFFFFFFFFFFFFF)
    for name, sc in (("uniform", uniform), ("bits", bits), ("all-ones-256", top)):
        d_s = torch.from_numpy(sc.view(np.int64)).cuda()
        got = np.zeros((1, 9), np.uint64)
        zk.check(L.zkt_secp_msm_dev(h, vp(d_s), n, None, ptr(got), None))
        tot = sum(k * s for k, s in zip(K, _ints(sc))) % SECP_N
        want = secp_arr([py_secp_mul(SECP_GEN, tot)])
        assert (got == want).all(), f"resident 2^17+1 secp256k1 MSM, {name} scalars"
    L.zkt_secp_bases_free(h)


def test_config3_tate_2p16(L):
    import torch
    n, half = 1 << 16, 1 << 15
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    a = rand_scalars(5, half); b = rand_scalars(6, half)
    kp, kq = np.concatenate([a, b]), np.concatenate([b, a])                      # element i: (a_i G1, b_i G2); element i + n/2: (b_i G1, a_i G2)
    d_p = torch.empty((n, G1W), dtype=torch.int64, device="cuda"); d_q = torch.empty((n, G2W), dtype=torch.int64, device="cuda")
    d_g1, d_kp = torch.from_numpy(_gen_rows(_g1_gen(), n).view(np.int64)).cuda(), torch.from_numpy(kp.view(np.int64)).cuda()
    d_g2, d_kq = torch.from_numpy(_gen_rows(_g2_gen(), n).view(np.int64)).cuda(), torch.from_numpy(kq.view(np.int64)).cuda()
    zk.check(L.zkt_g1_mul_batch_dev(vp(d_g1), vp(d_kp), 4, vp(d_p), n, None))
    zk.check(L.zkt_g2_mul_batch_dev(vp(d_g2), vp(d_kq), 4, vp(d_q), n, None))
    torch.cuda.synchronize()
    d_e = torch.empty((n, FQ12), dtype=torch.int64, device="cuda")
    zk.check(L.zkt_tate_batch_dev(vp(d_p), vp(d_q), vp(d_e), n, None))
    torch.cuda.synchronize()
    e = d_e.cpu().numpy().view(np.uint64)
    # bilinearity over the whole batch (pairing.rs:125-151): e(a G1, b G2) = e(G1,G2)^(ab) = e(b G1, a G2), different inputs, same bits
    assert (e[:half] == e[half:]).all()
    assert len({r.tobytes() for r in e[:half]}) == half, "degenerate outputs"
    # a 64-element sample, spread over the batch, against the oracle's reference algorithm (pairing.rs:86-100)
    idx = np.linspace(0, n - 1, 64).astype(np.int64)
    P = d_p.cpu().numpy().view(np.uint64)[idx].copy(); Qs = d_q.cpu().numpy().view(np.uint64)[idx].copy()
    want = np.zeros((64, FQ12), np.uint64)
    assert O.zkto_pairing_batch(3, ptr(P), ptr(Qs), ptr(want), 64, 16, None) == 0
    assert (e[idx] == want).all()
    # the host-pointer entry point on the same inputs gives the same bits
    got = np.zeros((256, FQ12), np.uint64)
    zk.check(L.zkt_tate_batch(ptr(d_p.cpu().numpy().view(np.uint64)[:256].copy()), ptr(d_q.cpu().numpy().view(np.uint64)[:256].copy()), ptr(got), 256))
    assert (got == e[:256]).all()
    # e(P_i, Q_i) e(-P_i, Q_i) == 1 for every element of a 2^16 batch (the fused verification kernels at full size)
    Pn = d_p.cpu().numpy().view(np.uint64); Qn = d_q.cpu().numpy().view(np.uint64)
    g1s = np.repeat(Pn, 2, axis=0); g2s = np.repeat(Qn, 2, axis=0)
    ok = np.zeros(n, np.uint32)
    zk.check(L.zkt_pairing_product_check_batch(ptr(g1s), ptr(g2s), (ctypes.c_uint8 * 2)(0, 1), 2, n, ok.ctypes.data))
    assert ok.all()
    g2s[2 * 777 + 1] = Qn[778]                                                   # one element with a mismatched pair: only that one fails
    zk.check(L.zkt_pairing_product_check_batch(ptr(g1s), ptr(g2s), (ctypes.c_uint8 * 2)(0, 1), 2, n, ok.ctypes.data))
    assert ok.sum() == n - 1 and ok[777] == 0


def test_config4_groth16_2p20(L):
    import torch
    n = 1 << 20
    mats, wires, l, m = chain_circuit_sparse(n, seed=7)
    rng = SplitMix64(7)
    fr = lambda x: ints_to_arr([x], 4)
    trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
    r, s = fr(rng.below(R - 1) + 1), fr(rng.below(R - 1) + 1)
    structs = [sparse_struct(*M) for M in mats]
    vk, vbuf = alloc_crs(1, l, m); vk.g1_uvw_wit = None
    pk = ctypes.c_void_p()
    zk.check(L.zkt_groth16_setup_r1cs(n, l, m, *[ctypes.addressof(x) for x in structs], *[t.ctypes.data for t in trap], ctypes.addressof(vk), ctypes.addressof(pk)))
    A, B, C = np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64)
    zk.check(L.zkt_groth16_prove_r1cs(pk, wires.ctypes.data, r.ctypes.data, s.ctypes.data, A.ctypes.data, B.ctypes.data, C.ctypes.data))
    # the proof POINTS at full size, bit for bit: their discrete logarithms from the injected trapdoor in O(n) python-integer arithmetic, times the generators in
    # python integers (qap_util.groth16_proof_scalars; neither the HIP path nor the oracle on the checking side)
    As, Bs, Cs = groth16_proof_scalars(mats, wires, l, trap, r, s)
    (gx1, gx0), (gy1, gy0) = G2_GEN
    assert (A == g1_arr([py_g1_mul(G1_GEN, As)])).all(), "A at 2^20 constraints"
    assert (B == g2_arr([to_abi_g2(py_g2_mul(((gx0, gx1), (gy0, gy1)), Bs))])).all(), "B at 2^20 constraints"
    assert (C == g1_arr([py_g1_mul(G1_GEN, Cs)])).all(), "C at 2^20 constraints"
    # the device-resident and the pipelined entry points give the same proof
    d_w = torch.from_numpy(wires.view(np.int64)).cuda()
    A2, B2, C2 = np.zeros_like(A), np.zeros_like(B), np.zeros_like(C)
    zk.check(L.zkt_groth16_prove_r1cs_submit(pk, 0, d_w.data_ptr(), r.ctypes.data, s.ctypes.data))
    zk.check(L.zkt_groth16_prove_r1cs_submit(pk, 1, d_w.data_ptr(), r.ctypes.data, s.ctypes.data))
    for slot in (0, 1):
        zk.check(L.zkt_groth16_prove_r1cs_collect(pk, slot, A2.ctypes.data, B2.ctypes.data, C2.ctypes.data))
        assert (A2 == A).all() and (B2 == B).all() and (C2 == C).all()
    # a witness that violates one constraint out of 2^20 must not verify
    bad_w = wires.copy(); bad_w[n // 2 + 3, 0] ^= np.uint64(2)
    zk.check(L.zkt_groth16_prove_r1cs(pk, bad_w.ctypes.data, r.ctypes.data, s.ctypes.data, A2.ctypes.data, B2.ctypes.data, C2.ctypes.data))
    L.zkt_groth16_pk_free(pk)
    stmt = wires[:l + 1].copy()
    assert L.zkt_groth16_verify(ctypes.byref(vk), ptr(A), ptr(B), ptr(C), ptr(stmt), l + 1) == 1          # verifier.rs:30-54
    bad_stmt = stmt.copy(); bad_stmt[l, 0] ^= np.uint64(1)
    assert L.zkt_groth16_verify(ctypes.byref(vk), ptr(A), ptr(B), ptr(C), ptr(bad_stmt), l + 1) == 0      # wrong statement
    assert L.zkt_groth16_verify(ctypes.byref(vk), ptr(A), ptr(B), ptr(A), ptr(stmt), l + 1) == 0          # tampered proof element
    assert L.zkt_groth16_verify(ctypes.byref(vk), ptr(A2), ptr(B2), ptr(C2), ptr(stmt), l + 1) == 0       # proof from the violating witness
    # BASELINE config 4 proper: ONE proof over 8 ranks.  The eight ranks' keys in turn on this card (index ranges of 2^17 / 2^18 terms: the sums run on the small-set
    # path, side by side), each rank's three Jacobian partials, then the combine step of the exchange — the unsharded proof bit for bit.
    W = 8
    parts = torch.zeros((W, zk.GROTH16_PARTIAL_WORDS), dtype=torch.int32, device="cuda")
    for k in range(W):
        vk2, vbuf2 = alloc_crs(1, l, m); vk2.g1_uvw_wit = None
        pk2 = ctypes.c_void_p()
        zk.check(L.zkt_groth16_setup_r1cs_sharded(n, l, m, *[ctypes.addressof(x) for x in structs], *[t.ctypes.data for t in trap], k, W, ctypes.addressof(vk2), ctypes.addressof(pk2)))
        for _ in range(2):                                                            # twice: the second proof replays the sums' graphs
            zk.check(L.zkt_groth16_prove_r1cs_partials(pk2, d_w.data_ptr(), r.ctypes.data, s.ctypes.data, parts[k].data_ptr()))
        L.zkt_groth16_pk_free(pk2)
    torch.cuda.synchronize()
    a, b = zk.G1_PARTIAL_WORDS, zk.G1_PARTIAL_WORDS + zk.G2_PARTIAL_WORDS
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    pa, pb, pc = parts[:, :a].contiguous(), parts[:, a:b].contiguous(), parts[:, b:].contiguous()
    zk.check(L.zkt_g1_jac_sum_dev(vp(pa), W, None, ptr(A2))); zk.check(L.zkt_g2_jac_sum_dev(vp(pb), W, None, ptr(B2))); zk.check(L.zkt_g1_jac_sum_dev(vp(pc), W, None, ptr(C2)))
    assert (A2 == A).all() and (B2 == B).all() and (C2 == C).all(), "proof sharded over 8 ranks differs from the unsharded proof"


def test_config5_range_proof_65536_bits(L):
    n = 1 << 16                                                                  # 64 bits x 1024 values as one 65,536-bit opening
    SG = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798, 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)
    g0 = np.zeros((1, 9), np.uint64); g0[0, :4] = int_to_limbs(SG[0], 4); g0[0, 4:8] = int_to_limbs(SG[1], 4)
    ks = rand_u64_array(11, (2 * n + 3, 4)); ks[:, 3] >>= np.uint64(1)
    pts = np.zeros((2 * n + 3, 9), np.uint64)
    zk.check(L.zkt_secp_mul_batch(ptr(np.repeat(g0, 2 * n + 3, axis=0)), ptr(ks), 4, ptr(pts), 2 * n + 3))
    gg, hh, u, g_r, h_r = pts[:n].copy(), pts[n:2 * n].copy(), pts[2 * n:2 * n + 1].copy(), pts[2 * n + 1:2 * n + 2].copy(), pts[2 * n + 2:].copy()
    bits = [int(v) for v in (rand_u64_array(15, (n,)) & np.uint64(1))]
    value = sum(bt << i for i, bt in enumerate(bits))
    aL = ints_to_arr(bits, 4)
    gamma = ints_to_arr([SplitMix64(17).below(SECP_N)], 4)
    tmp2, V = np.zeros((2, 9), np.uint64), np.zeros((1, 9), np.uint64)
    zk.check(L.zkt_secp_mul_batch(ptr(np.concatenate([g_r, h_r])), ptr(np.concatenate([ints_to_arr([value % SECP_N], 4), gamma])), 4, ptr(tmp2), 2))
    zk.check(L.zkt_secp_add_batch(ptr(tmp2[0:1].copy()), ptr(tmp2[1:2].copy()), ptr(V), 1))      # V = value g + gamma h (bulletproofs.rs:64)
    rnd = rand_u64_array(18, (7 + 2 * n, 4)); rnd[:, 3] >>= np.uint64(1); rnd[:, 0] |= np.uint64(1)
    xs = rand_u64_array(14, (16, 4)); xs[:, 3] >>= np.uint64(1); xs[:, 0] |= np.uint64(1)
    for use_ipa in (0, 1):
        assert L.zkt_bp_range_proof(n, ptr(V), ptr(aL), ptr(gamma), ptr(g_r), ptr(h_r), ptr(gg), ptr(hh), use_ipa, ptr(rnd), ptr(u), ptr(xs), None) == 1
    bad = aL.copy(); bad[40000, 0] ^= np.uint64(1)                               # a bit vector that does not open V
    for use_ipa in (0, 1):
        assert L.zkt_bp_range_proof(n, ptr(V), ptr(bad), ptr(gamma), ptr(g_r), ptr(h_r), ptr(gg), ptr(hh), use_ipa, ptr(rnd), ptr(u), ptr(xs), None) == 0
    # the same proof over a resident-generator context (the form bench.py times)
    rctx = ctypes.c_void_p(); zk.check(L.zkt_bp_ipa_ctx_create(n, ptr(gg), ptr(hh), ptr(u), ctypes.byref(rctx)))
    pts_a, pts_b = np.zeros((5, 9), np.uint64), np.zeros((5, 9), np.uint64)
    assert L.zkt_bp_range_proof(n, ptr(V), ptr(aL), ptr(gamma), ptr(g_r), ptr(h_r), ptr(gg), ptr(hh), 1, ptr(rnd), ptr(u), ptr(xs), ptr(pts_a)) == 1
    for use_ipa in (0, 1):
        assert L.zkt_bp_range_proof_ctx(rctx, ptr(V), ptr(aL), ptr(gamma), ptr(g_r), ptr(h_r), use_ipa, ptr(rnd), ptr(xs), ptr(pts_b)) == 1
        assert (pts_a == pts_b).all()                                            # A, S, T1, T2, P identical through both entry points
        assert L.zkt_bp_range_proof_ctx(rctx, ptr(V), ptr(bad), ptr(gamma), ptr(g_r), ptr(h_r), use_ipa, ptr(rnd), ptr(xs), None) == 0
    L.zkt_bp_ipa_ctx_free(rctx)
    # inner-product argument alone at 65,536 generators, resident generators: accept, and reject after one coefficient changes
    a, b = rand_u64_array(12, (n, 4)), rand_u64_array(13, (n, 4))
    a[:, 3] >>= np.uint64(1); b[:, 3] >>= np.uint64(1)
    c = sum(x * y for x, y in zip(_ints(a), _ints(b))) % SECP_N
    P = np.zeros((1, 9), np.uint64)
    zk.check(L.zkt_secp_msm(ptr(np.concatenate([gg, hh, u])), ptr(np.concatenate([a, b, ints_to_arr([c], 4)])), 2 * n + 1, ptr(P)))
    ctx = ctypes.c_void_p(); zk.check(L.zkt_bp_ipa_ctx_create(n, ptr(gg), ptr(hh), ptr(u), ctypes.byref(ctx)))
    assert L.zkt_bp_inner_product_argument_ctx(ctx, ptr(P), ptr(a), ptr(b), ptr(xs), None) == 1
    trace = np.zeros((16 * 3, 9), np.uint64)                                        # with a trace: the level-by-level form (32 L/R MSMs), same verdict
    assert L.zkt_bp_inner_product_argument_ctx(ctx, ptr(P), ptr(a), ptr(b), ptr(xs), ptr(trace)) == 1
    assert (trace[:, 8] == 0).all()
    a2 = a.copy(); a2[123, 0] ^= np.uint64(1)
    assert L.zkt_bp_inner_product_argument_ctx(ctx, ptr(P), ptr(a2), ptr(b), ptr(xs), None) == 0
    L.zkt_bp_ipa_ctx_free(ctx)


def test_verification_batches_at_full_size_decide_element_by_element(L):
    """The deciding entry points at the size where the one-element-per-lane 63-step kernels run (above the small-batch switch-over): 32,768 BLS signatures and 16,384
    Groth16 proofs, valid except at scattered positions that carry a wrong key, a swapped signature, a point outside G2 (left to the 255-step kernel), a tampered C, a wrong
    statement.  Exactly those positions must be rejected; the construction (sk * H(m) verifies under sk * g1, signature.rs:28-39; one proof replicated) is the checker here,
    the small parity tests pin the same kernels to the oracle in the forced child process."""
    from qap_util import example_cubic, qap_from_r1cs, dense, alloc_crs
    n = 1 << 15
    msgs = [b"m%06d" % i for i in range(n)]
    off = np.zeros(n + 1, np.uint64); off[1:] = np.cumsum([len(m) for m in msgs])
    buf = np.frombuffer(b"".join(msgs), dtype=np.uint8).copy()
    sks = rand_scalars(41, n, R); sks[sks.sum(axis=1) == 0] = 1
    pks = np.zeros((n, G1W), np.uint64); zk.check(L.zkt_bls_public_keys_batch(ptr(sks), n, ptr(pks)))
    sig = np.zeros((n, G2W), np.uint64); zk.check(L.zkt_bls_sign_batch(buf.ctypes.data, off.ctypes.data, ptr(sks), n, ptr(sig)))
    bad = {7: "key", 4099: "sig", 20000: "twist", n - 1: "key"}
    pks_t, sig_t = pks.copy(), sig.copy()
    for i, what in bad.items():
        if what == "key": pks_t[i] = pks[i - 1]
        elif what == "sig": sig_t[i] = sig[i + 1]
        else: sig_t[i] = g2_arr([to_abi_g2(py_twist_point(SplitMix64(9)))])[0]
    ok = np.zeros(n, np.uint32)
    zk.check(L.zkt_bls_verify_batch(buf.ctypes.data, off.ctypes.data, ptr(sig_t), ptr(pks_t), n, ok.ctypes.data))
    assert sorted(int(i) for i in np.nonzero(ok == 0)[0]) == sorted(bad)
    # Groth16: one valid proof of the reference's example, replicated
    A_, B_, C_, wit, l = example_cubic()
    nn, m = len(A_), len(wit) - 1
    ui, vi, wi, h, _ = qap_from_r1cs(A_, B_, C_, wit)
    U, V, W = dense(ui, nn), dense(vi, nn), dense(wi, nn)
    fr = lambda v: ints_to_arr([v], 4)
    rng = SplitMix64(171); trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
    crs, cbuf = alloc_crs(nn, l, m)
    zk.check(L.zkt_groth16_setup(ctypes.byref(crs), ptr(U), ptr(V), ptr(W), *[ptr(t) for t in trap]))
    pa, pb, pc = np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64)
    zk.check(L.zkt_groth16_prove(ctypes.byref(crs), ptr(U), ptr(V), ptr(ints_to_arr(wit, 4)), ptr(ints_to_arr(h, 4)), len(h), ptr(fr(12345)), ptr(fr(6789)), ptr(pa), ptr(pb), ptr(pc)))
    k = 1 << 14
    As, Bs, Cs = np.repeat(pa, k, axis=0), np.repeat(pb, k, axis=0), np.repeat(pc, k, axis=0)
    stmts = np.repeat(ints_to_arr(wit[:l + 1], 4).reshape(1, -1), k, axis=0).copy()
    badp = {3: "C", 5000: "stmt", 9999: "B-twist", k - 1: "stmt-big"}
    for i, what in badp.items():
        if what == "C": Cs[i] = pa[0]
        elif what == "stmt": stmts[i].reshape(l + 1, 4)[l, 0] ^= np.uint64(1)
        elif what == "B-twist": Bs[i] = g2_arr([to_abi_g2(py_twist_point(SplitMix64(10)))])[0]
        else: stmts[i].reshape(l + 1, 4)[l] = rand_scalars(43, 1, R)[0]             # a full-size field element: the statement tables' long path
    okp = np.zeros(k, np.uint32)
    zk.check(L.zkt_groth16_verify_batch(ctypes.byref(crs), ptr(As), ptr(Bs), ptr(Cs), ptr(stmts), l + 1, k, okp.ctypes.data))
    assert sorted(int(i) for i in np.nonzero(okp == 0)[0]) == sorted(badp)
