#!/usr/bin/env python3
"""Soak of the verifying-key cache under concurrency: T threads, K keys (more than the cache holds), every thread mixes single verifications (valid proof, wrong statement,
another key's proof), zkt_groth16_vk_prepare and batches of 4,096 on keys picked at random; every decision is checked.  usage: verify_soak.py [threads=8] [keys=6] [iterations=40]"""
import ctypes, importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from zkt_testlib import *
from qap_util import *
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8; K = int(sys.argv[2]) if len(sys.argv) > 2 else 6; ITER = int(sys.argv[3]) if len(sys.argv) > 3 else 40
zk = importlib.import_module("zk-toolkit_amd"); zk.init(); L = zk.lib()
fr = lambda v: ints_to_arr([v], 4)
A_, B_, C_, wit, l = example_cubic()
n, m = len(A_), len(wit) - 1
ui, vi, wi, h, _ = qap_from_r1cs(A_, B_, C_, wit)
U, V, W = dense(ui, n), dense(vi, n), dense(wi, n)
wires, H = ints_to_arr(wit, 4), ints_to_arr(h, 4)
stmt = ints_to_arr(wit[:l + 1], 4); bad = stmt.copy(); bad[l, 0] ^= np.uint64(1)
keys = []
for k in range(K):
    rng = SplitMix64(31000 + k); trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
    crs, buf = alloc_crs(n, l, m)
    zk.check(L.zkt_groth16_setup(ctypes.byref(crs), ptr(U), ptr(V), ptr(W), *[ptr(t) for t in trap]))
    pf = (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
    zk.check(L.zkt_groth16_prove(ctypes.byref(crs), ptr(U), ptr(V), ptr(wires), ptr(H), len(h), ptr(fr(rng.below(R - 1) + 1)), ptr(fr(rng.below(R - 1) + 1)), *[ptr(x) for x in pf]))
    keys.append((crs, buf, pf))
L.zkt_groth16_vk_prepare.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
errors, counts = [], [0] * T
start = threading.Barrier(T)
def worker(t):
    rng = SplitMix64(555 + t)
    try:
        start.wait()
        for it in range(ITER):
            k = rng.below(K); crs, pf = keys[k][0], keys[k][2]; other = keys[(k + 1) % K][2]
            what = rng.below(10)
            if what == 0:
                if L.zkt_groth16_vk_prepare(ctypes.byref(crs), l + 1) != 0: errors.append((t, it, "prepare"))
            elif what == 1:
                nb = 4096
                As, Bs, Cs = np.repeat(pf[0], nb, axis=0), np.repeat(pf[1], nb, axis=0), np.repeat(pf[2], nb, axis=0)
                sts = np.repeat(stmt.reshape(1, -1), nb, axis=0).copy(); j = rng.below(nb); sts[j] = bad.reshape(-1)
                ok = np.zeros(nb, np.uint32)
                rc = L.zkt_groth16_verify_batch(ctypes.byref(crs), ptr(As), ptr(Bs), ptr(Cs), ptr(sts), l + 1, nb, ok.ctypes.data)
                want = np.ones(nb, np.uint32); want[j] = 0
                if rc != 0 or not (ok == want).all(): errors.append((t, it, "batch", rc, int((ok != want).sum())))
            else:
                st, bd = stmt.copy(), bad.copy()
                got = (L.zkt_groth16_verify(ctypes.byref(crs), *[ptr(x) for x in pf], ptr(st), l + 1), L.zkt_groth16_verify(ctypes.byref(crs), *[ptr(x) for x in pf], ptr(bd), l + 1),
                       L.zkt_groth16_verify(ctypes.byref(crs), *[ptr(x) for x in other], ptr(st), l + 1))
                if got != (1, 0, 0): errors.append((t, it, k, got))
            counts[t] += 1
    except Exception as e:
        errors.append((t, repr(e)))
t0 = time.perf_counter()
ts = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
for t in ts: t.start()
for t in ts: t.join()
print("verify soak: %d threads x %d iterations on %d keys in %.1f s, %d operations, %d errors" % (T, ITER, K, time.perf_counter() - t0, sum(counts), len(errors)))
for e in errors[:10]: print("  ", e)
zk.shutdown() if hasattr(zk, "shutdown") else None
sys.exit(1 if errors else 0)
