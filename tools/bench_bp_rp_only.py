#!/usr/bin/env python3
"""Profiling helper: a few 65,536-bit range proofs over resident generators (zkt_bp_range_proof_ctx), nothing else — for rocprofv3 --kernel-trace."""
import ctypes, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from zkt_testlib import *
zk = importlib.import_module("zk-toolkit_amd"); zk.init(); L = zk.lib()
n = 1 << 16; use_ipa = int(sys.argv[1]) if len(sys.argv) > 1 else 0
SG = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798, 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)
g0 = np.zeros((1, 9), np.uint64); g0[0, :4] = int_to_limbs(SG[0], 4); g0[0, 4:8] = int_to_limbs(SG[1], 4)
ks = rand_u64_array(11, (2 * n + 3, 4)); ks[:, 3] >>= np.uint64(1)
pts = np.zeros((2 * n + 3, 9), np.uint64)
zk.check(L.zkt_secp_mul_batch(ptr(np.repeat(g0, 2 * n + 3, axis=0)), ptr(ks), 4, ptr(pts), 2 * n + 3))
gg, hh, u, g_r, h_r = pts[:n].copy(), pts[n:2 * n].copy(), pts[2 * n:2 * n + 1].copy(), pts[2 * n + 1:2 * n + 2].copy(), pts[2 * n + 2:].copy()
xs = rand_u64_array(14, (16, 4)); xs[:, 3] >>= np.uint64(1); xs[:, 0] |= np.uint64(1)
bits = [int(v) for v in (rand_u64_array(15, (n,)) & np.uint64(1))]
value = sum(bt << i for i, bt in enumerate(bits)); aL = ints_to_arr(bits, 4); gamma = ints_to_arr([SplitMix64(17).below(SECP_N)], 4)
tmp2, V = np.zeros((2, 9), np.uint64), np.zeros((1, 9), np.uint64)
zk.check(L.zkt_secp_mul_batch(ptr(np.concatenate([g_r, h_r])), ptr(np.concatenate([ints_to_arr([value % SECP_N], 4), gamma])), 4, ptr(tmp2), 2))
zk.check(L.zkt_secp_add_batch(ptr(tmp2[0:1].copy()), ptr(tmp2[1:2].copy()), ptr(V), 1))
rnd = rand_u64_array(18, (7 + 2 * n, 4)); rnd[:, 3] >>= np.uint64(1); rnd[:, 0] |= np.uint64(1)
ctx = ctypes.c_void_p(); zk.check(L.zkt_bp_ipa_ctx_create(n, ptr(gg), ptr(hh), ptr(u), ctypes.byref(ctx)))
for _ in range(4):
    t0 = time.perf_counter(); assert L.zkt_bp_range_proof_ctx(ctx, ptr(V), ptr(aL), ptr(gamma), ptr(g_r), ptr(h_r), use_ipa, ptr(rnd), ptr(xs), None) == 1
    print("range proof use_ipa=%d: %.2f ms" % (use_ipa, (time.perf_counter() - t0) * 1e3), flush=True)
L.zkt_bp_ipa_ctx_free(ctx)
