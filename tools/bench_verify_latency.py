import ctypes, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from zkt_testlib import *
from qap_util import *
zk = importlib.import_module("zk-toolkit_amd"); zk.init(); L = zk.lib()
fr = lambda v: ints_to_arr([v], 4)
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
stmt = ints_to_arr(wit[:l + 1], 4)
# one proof at a time against a key the library has never seen: calls 1, 2, 3.. (first sight is served by the 127-step kernels while the key's entry is built beside it: side stream +
# the key's two pairings behind the call's result; the second call queues behind those pairings, from the third on the 63-step loop at its steady latency)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); assert L.zkt_groth16_verify(ctypes.byref(crs), ptr(pa), ptr(pb), ptr(pc), ptr(stmt), l + 1) == 1; ts.append((time.perf_counter() - t0) * 1e3)
print("unprepared key, single verification, calls 1..5 (ms): " + " ".join("%.2f" % t for t in ts))
# a second key (other trapdoors), prepared explicitly before its first verification
sm2 = SplitMix64(72); trap2 = [fr(sm2.below(R - 1) + 1) for _ in range(5)]
crs2, buf2 = alloc_crs(nn, l, m)
zk.check(L.zkt_groth16_setup(ctypes.byref(crs2), ptr(U), ptr(V), ptr(W), *[ptr(t) for t in trap2]))
qa, qb, qc = np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64)
zk.check(L.zkt_groth16_prove(ctypes.byref(crs2), ptr(U), ptr(V), ptr(wires), ptr(H), len(h), ptr(fr(12345)), ptr(fr(6789)), ptr(qa), ptr(qb), ptr(qc)))
L.zkt_groth16_vk_prepare.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
t0 = time.perf_counter(); zk.check(L.zkt_groth16_vk_prepare(ctypes.byref(crs2), l + 1)); tp = (time.perf_counter() - t0) * 1e3
ts = []
for _ in range(5):
    t0 = time.perf_counter(); assert L.zkt_groth16_verify(ctypes.byref(crs2), ptr(qa), ptr(qb), ptr(qc), ptr(stmt), l + 1) == 1; ts.append((time.perf_counter() - t0) * 1e3)
print("prepared key (zkt_groth16_vk_prepare %.2f ms), single verification, calls 1..5 (ms): " % tp + " ".join("%.2f" % t for t in ts))
# a third key, unprepared again: what the first key's sequence paid once per process (side stream, pinned landing zone, events) is gone
sm3 = SplitMix64(73); trap3 = [fr(sm3.below(R - 1) + 1) for _ in range(5)]
crs3, buf3 = alloc_crs(nn, l, m)
zk.check(L.zkt_groth16_setup(ctypes.byref(crs3), ptr(U), ptr(V), ptr(W), *[ptr(t) for t in trap3]))
ra, rb, rc = np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64)
zk.check(L.zkt_groth16_prove(ctypes.byref(crs3), ptr(U), ptr(V), ptr(wires), ptr(H), len(h), ptr(fr(12345)), ptr(fr(6789)), ptr(ra), ptr(rb), ptr(rc)))
ts = []
for _ in range(5):
    t0 = time.perf_counter(); assert L.zkt_groth16_verify(ctypes.byref(crs3), ptr(ra), ptr(rb), ptr(rc), ptr(stmt), l + 1) == 1; ts.append((time.perf_counter() - t0) * 1e3)
print("another unprepared key, single verification, calls 1..5 (ms): " + " ".join("%.2f" % t for t in ts))
for k in (1, 16, 256, 1024, 4096):
    As, Bs, Cs = np.repeat(pa, k, axis=0), np.repeat(pb, k, axis=0), np.repeat(pc, k, axis=0)
    st = np.repeat(stmt.reshape(1, -1), k, axis=0).copy(); ok = np.zeros(k, np.uint32)
    for _ in range(2):      # time the steady state
        zk.check(L.zkt_groth16_verify_batch(ctypes.byref(crs), ptr(As), ptr(Bs), ptr(Cs), ptr(st), l + 1, k, ok.ctypes.data))
    t0 = time.perf_counter(); zk.check(L.zkt_groth16_verify_batch(ctypes.byref(crs), ptr(As), ptr(Bs), ptr(Cs), ptr(st), l + 1, k, ok.ctypes.data)); dt = time.perf_counter() - t0
    print("groth16 verify  proofs=%5d  %.2f ms  %.0f/s  all ok=%s  (ZKT_DPRODUCT_MAX=%s)" % (k, dt * 1e3, k / dt, bool(ok.all()), os.environ.get("ZKT_DPRODUCT_MAX", "default")))
