#!/usr/bin/env python3
"""Latency of Pinocchio setup / prove / verify on the reference's chain circuit; --reps verifications are timed one by one."""
import argparse, ctypes, importlib, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from zkt_testlib import *
from qap_util import *
ap = argparse.ArgumentParser()
ap.add_argument("--constraints", type=int, default=32)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
zk = importlib.import_module("zk-toolkit_amd"); zk.init(); L = zk.lib()
fr = lambda v: ints_to_arr([v], 4)
A_, B_, C_, wit, l = chain_circuit(args.constraints)
nn, n_io = len(A_), l + 1
Vp, Wp, Yp, hq, max_degree = pinocchio_instance(A_, B_, C_, wit)
sm = SplitMix64(5)
rnd = ints_to_arr([sm.below(R - 1) + 1 for _ in range(8)], 4)
crs, cbuf = alloc_pinocchio(nn, n_io, len(wit) - n_io, max_degree)
t0 = time.perf_counter(); zk.check(L.zkt_pinocchio_setup(ctypes.byref(crs), ptr(Vp), ptr(Wp), ptr(Yp), ptr(rnd))); t_s = time.perf_counter() - t0
pf, pbuf = alloc_pinocchio_proof()
wires, Hq = ints_to_arr(wit, 4), ints_to_arr(hq, 4)
t0 = time.perf_counter(); zk.check(L.zkt_pinocchio_prove(ctypes.byref(crs), ptr(wires), ptr(Hq), len(hq), ptr(fr(777)), ptr(fr(888)), ctypes.byref(pf))); t_p = time.perf_counter() - t0
io = wires[:n_io].copy()
tv = []
for _ in range(args.reps):
    t0 = time.perf_counter(); ok = L.zkt_pinocchio_verify(ctypes.byref(crs), ctypes.byref(pf), ptr(io)); tv.append(time.perf_counter() - t0)
    assert ok == 1
bad = io.copy(); bad[1, 0] ^= np.uint64(1)
assert L.zkt_pinocchio_verify(ctypes.byref(crs), ctypes.byref(pf), ptr(bad)) == 0
print(f"pinocchio n={nn} n_io={n_io}: setup {t_s*1e3:.1f} ms, prove {t_p*1e3:.1f} ms, verify " + " ".join(f"{t*1e3:.1f}" for t in tv) + " ms; wrong statement rejected")
