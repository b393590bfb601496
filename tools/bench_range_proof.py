import ctypes, importlib, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from zkt_testlib import *
zk = importlib.import_module("zk-toolkit_amd"); zk.init(); L = zk.lib()
n = 1 << 16
rng = SplitMix64(5)
SG = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798, 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)
g0 = np.zeros((1, 9), np.uint64); g0[0, :4] = int_to_limbs(SG[0], 4); g0[0, 4:8] = int_to_limbs(SG[1], 4)
ks = ints_to_arr([rng.below(SECP_N - 1) + 1 for _ in range(2 * n + 3)], 4)
pts = np.zeros((2 * n + 3, 9), np.uint64)
zk.check(L.zkt_secp_mul_batch(ptr(np.repeat(g0, 2 * n + 3, axis=0)), ptr(ks), 4, ptr(pts), 2 * n + 3))
gg, hh, g, h, u = pts[:n].copy(), pts[n:2 * n].copy(), pts[2 * n:2 * n + 1].copy(), pts[2 * n + 1:2 * n + 2].copy(), pts[2 * n + 2:].copy()
bits = [rng.below(2) for _ in range(n)]
value = sum(b << i for i, b in enumerate(bits))
aL = ints_to_arr(bits, 4)
gamma = ints_to_arr([rng.below(SECP_N - 1) + 1], 4)
tmp = np.zeros((2, 9), np.uint64); V = np.zeros((1, 9), np.uint64)
zk.check(L.zkt_secp_mul_batch(ptr(np.concatenate([g, h])), ptr(np.concatenate([ints_to_arr([value % SECP_N], 4), gamma])), 4, ptr(tmp), 2))
zk.check(L.zkt_secp_add_batch(ptr(tmp[0:1].copy()), ptr(tmp[1:2].copy()), ptr(V), 1))
rnd = ints_to_arr([rng.below(SECP_N - 1) + 1 for _ in range(7 + 2 * n)], 4)
xs = ints_to_arr([rng.below(SECP_N - 1) + 1 for _ in range(16)], 4)
for use_ipa in (0, 1):
    for rep in range(2):
        t0 = time.perf_counter()
        r = L.zkt_bp_range_proof(n, ptr(V), ptr(aL), ptr(gamma), ptr(g), ptr(h), ptr(gg), ptr(hh), use_ipa, ptr(rnd), ptr(u), ptr(xs), None)
        print("use_ipa", use_ipa, "->", r, f"{time.perf_counter()-t0:.4f}s", flush=True)
bad = aL.copy(); bad[7, 0] ^= 1
print("tampered ->", L.zkt_bp_range_proof(n, ptr(V), ptr(bad), ptr(gamma), ptr(g), ptr(h), ptr(gg), ptr(hh), 1, ptr(rnd), ptr(u), ptr(xs), None))
