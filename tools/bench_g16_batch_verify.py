#!/usr/bin/env python3
"""Groth16 batch verification throughput with small and with full-width statement values: 2^16 copies of one proof of the
reference's cubic example (statement 1, 35) and of a 4-constraint chain circuit (statement 1, w_4 — a 255-bit field element)."""
import ctypes, importlib, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from zkt_testlib import *
from qap_util import *
zk = importlib.import_module("zk-toolkit_amd"); zk.init(); L = zk.lib()
fr = lambda v: ints_to_arr([v], 4)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
for name, circ in (("cubic (statement 1, 35)", example_cubic()), ("chain4 (statement 1, 255-bit)", chain_circuit(4))):
    A_, B_, C_, wit, l = circ
    nn, m = len(A_), len(wit) - 1
    ui, vi, wi, h, _ = qap_from_r1cs(A_, B_, C_, wit)
    U, V, W = dense(ui, nn), dense(vi, nn), dense(wi, nn)
    sm = SplitMix64(71); trap = [fr(sm.below(R - 1) + 1) for _ in range(5)]
    crs, buf = alloc_crs(nn, l, m)
    zk.check(L.zkt_groth16_setup(ctypes.byref(crs), ptr(U), ptr(V), ptr(W), *[ptr(t) for t in trap]))
    wires = ints_to_arr(wit, 4); H = ints_to_arr(h, 4)
    pa, pb, pc = np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64)
    zk.check(L.zkt_groth16_prove(ctypes.byref(crs), ptr(U), ptr(V), ptr(wires), ptr(H), len(h), ptr(fr(12345)), ptr(fr(6789)), ptr(pa), ptr(pb), ptr(pc)))
    As, Bs, Cs = np.repeat(pa, k, axis=0), np.repeat(pb, k, axis=0), np.repeat(pc, k, axis=0)
    stmts = np.repeat(ints_to_arr(wit[:l + 1], 4).reshape(1, -1), k, axis=0).copy()
    stmts[7, 4] ^= np.uint64(1)                       # one wrong statement in the batch
    okv = np.zeros(k, np.uint32)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); zk.check(L.zkt_groth16_verify_batch(ctypes.byref(crs), ptr(As), ptr(Bs), ptr(Cs), ptr(stmts), l + 1, k, okv.ctypes.data)); ts.append(time.perf_counter() - t0)
    assert okv[7] == 0 and okv.sum() == k - 1
    print(f"{name}: {k} proofs, {min(ts)*1e3:.1f} ms = {k/min(ts)/1e3:.0f} k verifications/s (runs {' '.join('%.1f' % (t*1e3) for t in ts)} ms)", flush=True)
