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
# the prover with the evaluation key resident (zkt_pinocchio_pk): same proof points, no per-call upload of the ten base sets
pk = ctypes.c_void_p()
t0 = time.perf_counter(); zk.check(L.zkt_pinocchio_pk_create(ctypes.byref(crs), ctypes.byref(pk))); t_k = time.perf_counter() - t0
pf2, pbuf2 = alloc_pinocchio_proof()
tp = []
for _ in range(args.reps):
    t0 = time.perf_counter(); zk.check(L.zkt_pinocchio_prove_resident(pk, ptr(wires), ptr(Hq), len(hq), ptr(fr(777)), ptr(fr(888)), ctypes.byref(pf2))); tp.append(time.perf_counter() - t0)
assert all((pbuf[k] == pbuf2[k]).all() for k in pbuf)
L.zkt_pinocchio_pk_free(pk)
print(f"  resident key: create {t_k*1e3:.1f} ms, prove " + " ".join(f"{t*1e3:.2f}" for t in tp) + " ms (same nine points)")
# the prover at 2^16 mid wires on a synthetic evaluation key (bases = known generator multiples; tests/test_gpu_pinocchio.py checks such a proof by linearity)
n_mid, n_io2, deg = 1 << 16, 3, 1 << 16
gen = np.random.Generator(np.random.PCG64(31))
def scal(cnt):
    a = gen.integers(0, 2**63, size=(cnt, 4), dtype=np.uint64); a[:, 3] >>= np.uint64(2); return a
crs2, buf2 = alloc_pinocchio(n_mid, n_io2, n_mid, deg)
g2 = np.zeros((1, G2W), np.uint64); oracle().zkto_g2_generator(ptr(g2))
for name, w, c in qap_util_pin_ek() + qap_util_pin_vk():
    cnt = buf2[name].shape[0]; k = scal(cnt)
    if w == G1W: zk.check(L.zkt_bls_public_keys_batch(ptr(k), cnt, ptr(buf2[name])))
    else: zk.check(L.zkt_g2_mul_batch(ptr(np.repeat(g2, cnt, axis=0)), ptr(k), 4, ptr(buf2[name]), cnt))
w2, H2 = scal(n_io2 + n_mid), scal(deg)
pk = ctypes.c_void_p()
t0 = time.perf_counter(); zk.check(L.zkt_pinocchio_pk_create(ctypes.byref(crs2), ctypes.byref(pk))); t_k = time.perf_counter() - t0
pf3, pbuf3 = alloc_pinocchio_proof()
tp = []
for _ in range(args.reps):
    t0 = time.perf_counter(); zk.check(L.zkt_pinocchio_prove_resident(pk, ptr(w2), ptr(H2), deg, ptr(fr(777)), ptr(fr(888)), ctypes.byref(pf3))); tp.append(time.perf_counter() - t0)
L.zkt_pinocchio_pk_free(pk)
pf4, pbuf4 = alloc_pinocchio_proof()
t0 = time.perf_counter(); zk.check(L.zkt_pinocchio_prove(ctypes.byref(crs2), ptr(w2), ptr(H2), deg, ptr(fr(777)), ptr(fr(888)), ctypes.byref(pf4))); t_one = time.perf_counter() - t0
assert all((pbuf3[k] == pbuf4[k]).all() for k in pbuf3)
print(f"pinocchio prover, 2^16 mid wires, quotient degree 2^16: resident key create {t_k*1e3:.0f} ms, prove " + " ".join(f"{t*1e3:.1f}" for t in tp) + f" ms; one-shot entry point {t_one*1e3:.0f} ms (same nine points)")
