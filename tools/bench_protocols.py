#!/usr/bin/env python3
"""Throughput of the protocol-level entry points on one MI355X, host-pointer API (PCIe included).
Prints one JSON object; kept as profiles/r02_protocols.json."""
import ctypes, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from zkt_testlib import *
from qap_util import *
from concurrent.futures import ThreadPoolExecutor
zk = importlib.import_module("zk-toolkit_amd"); zk.init(); L = zk.lib()
O = oracle()                      # CPU baselines only: the oracle restates the reference algorithm (checker / baseline, never the product path)
CORES = min(os.cpu_count() or 1, 16)
fr = lambda v: ints_to_arr([v], 4)
res = {}


def g1_gen():
    g = np.zeros((1, G1W), np.uint64); g[0, :6] = int_to_limbs(G1_GEN[0], 6); g[0, 6:12] = int_to_limbs(G1_GEN[1], 6); return g


# ---- BLS signature verification (signature.rs:34-39), 2^16 signatures, 32-byte messages -------------------------------------
n = 1 << 16
rng = np.random.Generator(np.random.PCG64(5))
msgs = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
off = (np.arange(n + 1, dtype=np.uint64) * 32)
sks = rand_scalars(9, n)
pks = np.zeros((n, G1W), np.uint64); zk.check(L.zkt_bls_public_keys_batch(ptr(sks), n, ptr(pks)))
t0 = time.perf_counter(); zk.check(L.zkt_bls_public_keys_batch(ptr(sks), n, ptr(pks))); t_keys = time.perf_counter() - t0
sigs = np.zeros((n, G2W), np.uint64)
t0 = time.perf_counter(); zk.check(L.zkt_bls_sign_batch(msgs.ctypes.data, off.ctypes.data, ptr(sks), n, ptr(sigs))); t_sign = time.perf_counter() - t0
ok = np.zeros(n, np.uint32)
zk.check(L.zkt_bls_verify_batch(msgs.ctypes.data, off.ctypes.data, ptr(sigs), ptr(pks), n, ok.ctypes.data))
t0 = time.perf_counter(); zk.check(L.zkt_bls_verify_batch(msgs.ctypes.data, off.ctypes.data, ptr(sigs), ptr(pks), n, ok.ctypes.data)); t_ver = time.perf_counter() - t0
assert ok.all()
# CPU baseline: signature.rs:34-39 is two tate() calls (+ one G2 scalar multiplication for the hash); 2*CORES pairings on CORES threads
mc = CORES
g1r = np.repeat(g1_gen(), mc, axis=0)
Hc = np.zeros((mc, G2W), np.uint64)
g2g = np.zeros((1, G2W), np.uint64); O.zkto_g2_generator(ptr(g2g))
t0 = time.perf_counter()
assert O.zkto_g2_mul_batch(ptr(np.repeat(g2g, mc, axis=0)), ptr(ints_to_arr([int.from_bytes(bytes(m_), "big") % R for m_ in msgs[:mc]], 4)), 4, ptr(Hc), mc, CORES) == 0
e1, e2 = np.zeros((mc, FQ12), np.uint64), np.zeros((mc, FQ12), np.uint64)
assert O.zkto_pairing_batch(3, ptr(g1r), ptr(sigs[:mc].copy()), ptr(e1), mc, CORES, None) == 0
assert O.zkto_pairing_batch(3, ptr(pks[:mc].copy()), ptr(Hc), ptr(e2), mc, CORES, None) == 0
t_cpu = time.perf_counter() - t0
assert (e1 == e2).all()
res["bls"] = {"signatures": n, "public_keys_per_s": n / t_keys, "sign_per_s": n / t_sign, "verify_per_s": n / t_ver, "all_valid": True,
              "cpu_baseline": {"verify_per_s": mc / t_cpu, "cores": CORES, "kind": "port", "sample": "%d verifications by the oracle's reference algorithm, %.1f s" % (mc, t_cpu)}}

# ---- Groth16 batch verification (f-2): 2^16 proofs of the reference's example circuit against one CRS ----------------------------
A_, B_, C_, wit, l = example_cubic()
nn, m = len(A_), len(wit) - 1
ui, vi, wi, h, _ = qap_from_r1cs(A_, B_, C_, wit)
U, V, W = dense(ui, nn), dense(vi, nn), dense(wi, nn)
sm = SplitMix64(71); trap = [fr(sm.below(R - 1) + 1) for _ in range(5)]
crs, buf = alloc_crs(nn, l, m)
zk.check(L.zkt_groth16_setup(ctypes.byref(crs), ptr(U), ptr(V), ptr(W), *[ptr(t) for t in trap]))
wires = ints_to_arr(wit, 4); H = ints_to_arr(h, 4)
pa, pb, pc = np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64)
zk.check(L.zkt_groth16_prove(ctypes.byref(crs), ptr(U), ptr(V), ptr(wires), ptr(H), len(h), ptr(fr(12345)), ptr(fr(6789)), ptr(pa), ptr(pb), ptr(pc)))
k = 1 << 16
As, Bs, Cs = np.repeat(pa, k, axis=0), np.repeat(pb, k, axis=0), np.repeat(pc, k, axis=0)
stmts = np.repeat(ints_to_arr(wit[:l + 1], 4).reshape(1, -1), k, axis=0).copy()
okv = np.zeros(k, np.uint32)
zk.check(L.zkt_groth16_verify_batch(ctypes.byref(crs), ptr(As), ptr(Bs), ptr(Cs), ptr(stmts), l + 1, k, okv.ctypes.data))
t0 = time.perf_counter(); zk.check(L.zkt_groth16_verify_batch(ctypes.byref(crs), ptr(As), ptr(Bs), ptr(Cs), ptr(stmts), l + 1, k, okv.ctypes.data)); t_v = time.perf_counter() - t0
assert okv.all()
# the same batch with full-size statement values (public inputs are field elements: hashes, commitments): every proof is then rejected, the work is the same
# except for the statement sums, which no longer profit from short scalars
big = np.random.Generator(np.random.PCG64(5)).integers(0, 2**63, size=(k * (l + 1), 4), dtype=np.uint64); big[:, 3] >>= np.uint64(2)
zk.check(L.zkt_groth16_verify_batch(ctypes.byref(crs), ptr(As), ptr(Bs), ptr(Cs), ptr(big), l + 1, k, okv.ctypes.data))      # untimed first call, like the valid batch above
t0 = time.perf_counter(); zk.check(L.zkt_groth16_verify_batch(ctypes.byref(crs), ptr(As), ptr(Bs), ptr(Cs), ptr(big), l + 1, k, okv.ctypes.data)); t_vb = time.perf_counter() - t0
assert not okv.any()
ocrs = Crs(n=nn, l=l, m=m)
for kf in buf: setattr(ocrs, kf, ptr(buf[kf]))
stmt1 = ints_to_arr(wit[:l + 1], 4)
def _one(_):
    return O.zkto_groth16_verify(ctypes.byref(ocrs), ptr(pa), ptr(pb), ptr(pc), ptr(stmt1), l + 1)
t0 = time.perf_counter()
with ThreadPoolExecutor(CORES) as ex: outs = list(ex.map(_one, range(CORES)))
t_cpu = time.perf_counter() - t0
assert all(o == 1 for o in outs)
res["groth16_verify_batch"] = {"proofs": k, "statement_wires": l + 1, "verifications_per_s": k / t_v, "verifications_per_s_full_size_statements": k / t_vb,
                               "cpu_baseline": {"verifications_per_s": CORES / t_cpu, "cores": CORES, "kind": "port",
                                                "sample": "%d verifications (3 tate() each, verifier.rs:30-54) by the oracle, one per thread, %.1f s" % (CORES, t_cpu)}}

# ---- Bulletproofs inner-product argument, n = 64*1024 generators, 16 levels (BASELINE config 5 shape) ---------------------------
n = 1 << 16
sm = SplitMix64(77)
O_g = np.zeros((1, 9), np.uint64)
SG = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798, 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)      # secp256k1/affine_point.rs:40-47
O_g[0, :4] = int_to_limbs(SG[0], 4); O_g[0, 4:8] = int_to_limbs(SG[1], 4)
ks = rand_scalars(11, 2 * n + 1, SECP_N)
pts = np.zeros((2 * n + 1, 9), np.uint64)
zk.check(L.zkt_secp_mul_batch(ptr(np.repeat(O_g, 2 * n + 1, axis=0)), ptr(ks), 4, ptr(pts), 2 * n + 1))
gg, hh, u = pts[:n].copy(), pts[n:2 * n].copy(), pts[2 * n:].copy()
a, b = rand_scalars(12, n, SECP_N), rand_scalars(13, n, SECP_N)
c = sum(limbs_to_int(x) * limbs_to_int(y) for x, y in zip(a, b)) % SECP_N
P = np.zeros((1, 9), np.uint64)
zk.check(L.zkt_secp_msm(ptr(np.concatenate([gg, hh, u])), ptr(np.concatenate([a, b, ints_to_arr([c], 4)])), 2 * n + 1, ptr(P)))
xs = rand_scalars(14, 16, SECP_N); xs[:, 0] |= np.uint64(1)
assert L.zkt_bp_inner_product_argument(n, ptr(gg), ptr(hh), ptr(u), ptr(P), ptr(a), ptr(b), ptr(xs), None) == 1
t0 = time.perf_counter(); r = L.zkt_bp_inner_product_argument(n, ptr(gg), ptr(hh), ptr(u), ptr(P), ptr(a), ptr(b), ptr(xs), None); t_ipa = time.perf_counter() - t0
assert r == 1
ctx = ctypes.c_void_p()
t0 = time.perf_counter(); zk.check(L.zkt_bp_ipa_ctx_create(n, ptr(gg), ptr(hh), ptr(u), ctypes.byref(ctx))); t_ipa_setup = time.perf_counter() - t0
assert L.zkt_bp_inner_product_argument_ctx(ctx, ptr(P), ptr(a), ptr(b), ptr(xs), None) == 1
t0 = time.perf_counter()
for _ in range(4): assert L.zkt_bp_inner_product_argument_ctx(ctx, ptr(P), ptr(a), ptr(b), ptr(xs), None) == 1
t_ipa_res = (time.perf_counter() - t0) / 4
L.zkt_bp_ipa_ctx_free(ctx)
nc = 256                           # CPU baseline at 256 generators (8 levels); the reference's cost is linear in the number of generators
ot = np.zeros((8 * 3, 9), np.uint64)
Pc = np.zeros((1, 9), np.uint64); assert O.zkto_bp_commit(nc, ptr(gg[:nc].copy()), ptr(hh[:nc].copy()), ptr(u), ptr(a[:nc].copy()), ptr(b[:nc].copy()), ptr(Pc)) == 0
t0 = time.perf_counter()
assert O.zkto_bp_ipa(nc, ptr(gg[:nc].copy()), ptr(hh[:nc].copy()), ptr(u), ptr(Pc), ptr(a[:nc].copy()), ptr(b[:nc].copy()), ptr(xs[:8].copy()), ptr(ot)) == 1
t_cpu = time.perf_counter() - t0
res["bulletproofs_ipa"] = {"generators": n, "levels": 16, "seconds": t_ipa, "seconds_resident_generators": t_ipa_res, "generator_setup_seconds": t_ipa_setup, "accepts": True,
                           "cpu_baseline": {"generators": nc, "seconds": t_cpu, "cores": 1, "kind": "port",
                                            "sample": "oracle's reference algorithm at %d generators; x%d for %d generators = %.0f s" % (nc, n // nc, n, t_cpu * n / nc)}}

# ---- Bulletproofs range proof (a18, BASELINE config 5 shape: 64 bits x 1024 values = one 65,536-bit opening) ------------------
rbits = [int(v) for v in (rand_scalars(15, n, 2)[:, 0] & np.uint64(1))]
rvalue = sum(bt << i for i, bt in enumerate(rbits))
extra = np.zeros((2, 9), np.uint64)
zk.check(L.zkt_secp_mul_batch(ptr(np.repeat(O_g, 2, axis=0)), ptr(rand_scalars(16, 2, SECP_N)), 4, ptr(extra), 2))
g_r, h_r = extra[0:1].copy(), extra[1:2].copy()
aL = ints_to_arr(rbits, 4); gamma = rand_scalars(17, 1, SECP_N)
tmp2 = np.zeros((2, 9), np.uint64); V = np.zeros((1, 9), np.uint64)
zk.check(L.zkt_secp_mul_batch(ptr(np.concatenate([g_r, h_r])), ptr(np.concatenate([ints_to_arr([rvalue % SECP_N], 4), gamma])), 4, ptr(tmp2), 2))
zk.check(L.zkt_secp_add_batch(ptr(tmp2[0:1].copy()), ptr(tmp2[1:2].copy()), ptr(V), 1))
rnd_r = rand_scalars(18, 7 + 2 * n, SECP_N); rnd_r[:, 0] |= np.uint64(1)
t_rp = {}
for use_ipa in (0, 1):
    assert L.zkt_bp_range_proof(n, ptr(V), ptr(aL), ptr(gamma), ptr(g_r), ptr(h_r), ptr(gg), ptr(hh), use_ipa, ptr(rnd_r), ptr(u), ptr(xs), None) == 1
    t0 = time.perf_counter()
    assert L.zkt_bp_range_proof(n, ptr(V), ptr(aL), ptr(gamma), ptr(g_r), ptr(h_r), ptr(gg), ptr(hh), use_ipa, ptr(rnd_r), ptr(u), ptr(xs), None) == 1
    t_rp[use_ipa] = time.perf_counter() - t0
bad = aL.copy(); bad[7, 0] ^= np.uint64(1)
assert L.zkt_bp_range_proof(n, ptr(V), ptr(bad), ptr(gamma), ptr(g_r), ptr(h_r), ptr(gg), ptr(hh), 1, ptr(rnd_r), ptr(u), ptr(xs), None) == 0
res["bulletproofs_range_proof"] = {"bits": n, "seconds_without_ipa": t_rp[0], "seconds_with_ipa": t_rp[1], "accepts": True, "rejects_wrong_opening": True,
                                   "note": "Bulletproofs::range_proof (bulletproofs.rs:58-147), every random draw injected; host-pointer API, PCIe included"}

# ---- Pinocchio (f-4): chain circuit with 32 constraints ---------------------------------------------------------------------
A_, B_, C_, wit, l = chain_circuit(32)
nn, n_io = len(A_), l + 1
Vp, Wp, Yp, hq, max_degree = pinocchio_instance(A_, B_, C_, wit)
sm = SplitMix64(5)
rnd = ints_to_arr([sm.below(R - 1) + 1 for _ in range(8)], 4)
crs, cbuf = alloc_pinocchio(nn, n_io, len(wit) - n_io, max_degree)
t0 = time.perf_counter(); zk.check(L.zkt_pinocchio_setup(ctypes.byref(crs), ptr(Vp), ptr(Wp), ptr(Yp), ptr(rnd))); t_s = time.perf_counter() - t0
pf, pbuf = alloc_pinocchio_proof()
wires, Hq = ints_to_arr(wit, 4), ints_to_arr(hq, 4)
t0 = time.perf_counter(); zk.check(L.zkt_pinocchio_prove(ctypes.byref(crs), ptr(wires), ptr(Hq), len(hq), ptr(fr(777)), ptr(fr(888)), ctypes.byref(pf))); t_p = time.perf_counter() - t0
io_p = wires[:n_io].copy()
t0 = time.perf_counter(); okp = L.zkt_pinocchio_verify(ctypes.byref(crs), ctypes.byref(pf), ptr(io_p)); t_v0 = time.perf_counter() - t0
assert okp == 1
t_v = []
for _ in range(5):
    t0 = time.perf_counter(); okp = L.zkt_pinocchio_verify(ctypes.byref(crs), ctypes.byref(pf), ptr(io_p)); t_v.append(time.perf_counter() - t0)
    assert okp == 1
res["pinocchio"] = {"constraints": nn, "setup_s": t_s, "prove_s": t_p, "verify_first_call_s": t_v0, "verify_s": sorted(t_v)[len(t_v) // 2], "accepts": True,
                    "note": "latency of a single small proof, not a throughput figure; prove goes through the host-pointer one-shot MSM entry points; "
                            "verify_first_call_s includes building the fixed-base tables of the key's io points, verify_s is the median of five more calls on that key"}
print(json.dumps(res, indent=1))
