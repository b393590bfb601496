#!/usr/bin/env python3
"""Device memory retained by the queues after single Groth16 verifications (the runtime keeps a queue's scratch): free memory before the first
verification, after the first (unprepared key: guard + side + legacy queues at work), after five more, after a batch of 4,096 on the per-lane kernels."""
import ctypes, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from zkt_testlib import *
from qap_util import *
zk = importlib.import_module("zk-toolkit_amd"); zk.init(); L = zk.lib()
fr = lambda v: ints_to_arr([v], 4)
A_, B_, C_, wit, l = example_cubic(); n, m = len(A_), len(wit) - 1
ui, vi, wi, h, _ = qap_from_r1cs(A_, B_, C_, wit); U, V, W = dense(ui, n), dense(vi, n), dense(wi, n)
wires, H = ints_to_arr(wit, 4), ints_to_arr(h, 4); stmt = ints_to_arr(wit[:l + 1], 4)
sm = SplitMix64(71); trap = [fr(sm.below(R - 1) + 1) for _ in range(5)]
crs, buf = alloc_crs(n, l, m)
zk.check(L.zkt_groth16_setup(ctypes.byref(crs), ptr(U), ptr(V), ptr(W), *[ptr(t) for t in trap]))
pa, pb, pc = np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64)
zk.check(L.zkt_groth16_prove(ctypes.byref(crs), ptr(U), ptr(V), ptr(wires), ptr(H), len(h), ptr(fr(12345)), ptr(fr(6789)), ptr(pa), ptr(pb), ptr(pc)))
free = lambda: torch.cuda.mem_get_info()[0] / 2**30
torch.cuda.synchronize(); f0 = free(); print("free before the first verification: %.2f GiB" % f0)
assert L.zkt_groth16_verify(ctypes.byref(crs), ptr(pa), ptr(pb), ptr(pc), ptr(stmt), l + 1) == 1
torch.cuda.synchronize(); f1 = free(); print("after the first (unprepared key):   %.2f GiB  (-%.2f)" % (f1, f0 - f1))
for _ in range(5): assert L.zkt_groth16_verify(ctypes.byref(crs), ptr(pa), ptr(pb), ptr(pc), ptr(stmt), l + 1) == 1
torch.cuda.synchronize(); f2 = free(); print("after five more:                    %.2f GiB  (-%.2f)" % (f2, f0 - f2))
k = 32768
As, Bs, Cs = np.repeat(pa, k, axis=0), np.repeat(pb, k, axis=0), np.repeat(pc, k, axis=0); st = np.repeat(stmt.reshape(1, -1), k, axis=0).copy(); ok = np.zeros(k, np.uint32)
zk.check(L.zkt_groth16_verify_batch(ctypes.byref(crs), ptr(As), ptr(Bs), ptr(Cs), ptr(st), l + 1, k, ok.ctypes.data)); assert ok.all()
torch.cuda.synchronize(); f3 = free(); print("after a batch of 32,768:            %.2f GiB  (-%.2f)" % (f3, f0 - f3))
