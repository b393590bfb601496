#!/usr/bin/env python3
"""Bulletproofs at BASELINE config 5's size (65,536 generators): inner-product argument with resident generators and the range proof, timed."""
import ctypes, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from zkt_testlib import *
zk = importlib.import_module("zk-toolkit_amd"); zk.init(); L = zk.lib()
n = 1 << 16
SG = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798, 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)
g0 = np.zeros((1, 9), np.uint64); g0[0, :4] = int_to_limbs(SG[0], 4); g0[0, 4:8] = int_to_limbs(SG[1], 4)
ks = rand_u64_array(11, (2 * n + 3, 4)); ks[:, 3] >>= np.uint64(1)
pts = np.zeros((2 * n + 3, 9), np.uint64)
zk.check(L.zkt_secp_mul_batch(ptr(np.repeat(g0, 2 * n + 3, axis=0)), ptr(ks), 4, ptr(pts), 2 * n + 3))
gg, hh, u, g_r, h_r = pts[:n].copy(), pts[n:2 * n].copy(), pts[2 * n:2 * n + 1].copy(), pts[2 * n + 1:2 * n + 2].copy(), pts[2 * n + 2:].copy()
a, b = rand_u64_array(12, (n, 4)), rand_u64_array(13, (n, 4)); a[:, 3] >>= np.uint64(1); b[:, 3] >>= np.uint64(1)
ints = lambda x: [int.from_bytes(r.tobytes(), "little") for r in np.ascontiguousarray(x)]
c = sum(x * y for x, y in zip(ints(a), ints(b))) % SECP_N
P = np.zeros((1, 9), np.uint64)
zk.check(L.zkt_secp_msm(ptr(np.concatenate([gg, hh, u])), ptr(np.concatenate([a, b, ints_to_arr([c], 4)])), 2 * n + 1, ptr(P)))
xs = rand_u64_array(14, (16, 4)); xs[:, 3] >>= np.uint64(1); xs[:, 0] |= np.uint64(1)
ctx = ctypes.c_void_p(); zk.check(L.zkt_bp_ipa_ctx_create(n, ptr(gg), ptr(hh), ptr(u), ctypes.byref(ctx)))
assert L.zkt_bp_inner_product_argument_ctx(ctx, ptr(P), ptr(a), ptr(b), ptr(xs), None) == 1
ts = []
for _ in range(5):
    t0 = time.perf_counter(); assert L.zkt_bp_inner_product_argument_ctx(ctx, ptr(P), ptr(a), ptr(b), ptr(xs), None) == 1; ts.append(time.perf_counter() - t0)
print("IPA 65,536 generators resident: best %.2f ms median %.2f ms" % (min(ts) * 1e3, sorted(ts)[2] * 1e3))
L.zkt_bp_ipa_ctx_free(ctx)
bits = [int(v) for v in (rand_u64_array(15, (n,)) & np.uint64(1))]
value = sum(bt << i for i, bt in enumerate(bits)); aL = ints_to_arr(bits, 4); gamma = ints_to_arr([SplitMix64(17).below(SECP_N)], 4)
tmp2, V = np.zeros((2, 9), np.uint64), np.zeros((1, 9), np.uint64)
zk.check(L.zkt_secp_mul_batch(ptr(np.concatenate([g_r, h_r])), ptr(np.concatenate([ints_to_arr([value % SECP_N], 4), gamma])), 4, ptr(tmp2), 2))
zk.check(L.zkt_secp_add_batch(ptr(tmp2[0:1].copy()), ptr(tmp2[1:2].copy()), ptr(V), 1))
rnd = rand_u64_array(18, (7 + 2 * n, 4)); rnd[:, 3] >>= np.uint64(1); rnd[:, 0] |= np.uint64(1)
for use_ipa in (0, 1):
    assert L.zkt_bp_range_proof(n, ptr(V), ptr(aL), ptr(gamma), ptr(g_r), ptr(h_r), ptr(gg), ptr(hh), use_ipa, ptr(rnd), ptr(u), ptr(xs), None) == 1
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); assert L.zkt_bp_range_proof(n, ptr(V), ptr(aL), ptr(gamma), ptr(g_r), ptr(h_r), ptr(gg), ptr(hh), use_ipa, ptr(rnd), ptr(u), ptr(xs), None) == 1; ts.append(time.perf_counter() - t0)
    print("range proof 65,536 bits use_ipa=%d: best %.2f ms" % (use_ipa, min(ts) * 1e3))
ctx = ctypes.c_void_p(); zk.check(L.zkt_bp_ipa_ctx_create(n, ptr(gg), ptr(hh), ptr(u), ctypes.byref(ctx)))
for use_ipa in (0, 1):
    assert L.zkt_bp_range_proof_ctx(ctx, ptr(V), ptr(aL), ptr(gamma), ptr(g_r), ptr(h_r), use_ipa, ptr(rnd), ptr(xs), None) == 1
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); assert L.zkt_bp_range_proof_ctx(ctx, ptr(V), ptr(aL), ptr(gamma), ptr(g_r), ptr(h_r), use_ipa, ptr(rnd), ptr(xs), None) == 1; ts.append(time.perf_counter() - t0)
    print("range proof 65,536 bits, generators resident, use_ipa=%d: best %.2f ms" % (use_ipa, min(ts) * 1e3))
bad = aL.copy(); bad[777, 0] ^= np.uint64(1)
assert L.zkt_bp_range_proof_ctx(ctx, ptr(V), ptr(bad), ptr(gamma), ptr(g_r), ptr(h_r), 1, ptr(rnd), ptr(xs), None) == 0
L.zkt_bp_ipa_ctx_free(ctx)
