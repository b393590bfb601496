#!/usr/bin/env python3
"""Window width of the resident MSM as a function of the set size: one process per ZKT_MSM_C (a forced width; the plan reads it once; unset = the plan's own table), every size and group inside.
Prints single-MSM latency (one at a time) and the pipelined time per MSM (three in flight).  usage: msm_window_sweep.py <g1|g2> <log2n> [<log2n> ...]"""
import ctypes, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
from zkt_testlib import G1_GEN, G2_GEN, int_to_limbs
zk = importlib.import_module("zk-toolkit_amd"); zk.init(0); L = zk.lib()
grp = sys.argv[1]; sizes = [int(x) for x in sys.argv[2:]]
dev = torch.device("cuda", 0); sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream); vp = lambda t: ctypes.c_void_p(t.data_ptr())
W = 13 if grp == "g1" else 25
gen = np.zeros((1, W), np.uint64)
if grp == "g1": gen[0, :6] = int_to_limbs(G1_GEN[0], 6); gen[0, 6:12] = int_to_limbs(G1_GEN[1], 6)
else:
    (x1, x0), (y1, y0) = G2_GEN
    gen[0, 0:6] = int_to_limbs(x1, 6); gen[0, 6:12] = int_to_limbs(x0, 6); gen[0, 12:18] = int_to_limbs(y1, 6); gen[0, 18:24] = int_to_limbs(y0, 6)
F = lambda name: getattr(L, "zkt_%s_%s" % (grp, name))
for log2n in sizes:
    n = 1 << log2n
    d_g = torch.from_numpy(np.repeat(gen, n, axis=0).view(np.int64)).to(dev)
    d_k = torch.from_numpy(bench.rand_scalars_mod_r(3, n).view(np.int64)).to(dev)
    d_b = torch.empty((n, W), dtype=torch.int64, device=dev)
    zk.check(F("mul_batch_dev")(vp(d_g), vp(d_k), 4, vp(d_b), n, sp)); torch.cuda.synchronize()
    h = ctypes.c_void_p(); zk.check(F("bases_from_device")(vp(d_b), n, sp, ctypes.byref(h)))
    d_s = torch.from_numpy(bench.rand_scalars_mod_r(4, n).view(np.int64)).to(dev)
    out = np.zeros((1, W), np.uint64); op = out.ctypes.data_as(ctypes.c_void_p)
    for _ in range(3): zk.check(F("msm_dev")(h, vp(d_s), n, sp, op, None))
    ts = []
    for _ in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter(); zk.check(F("msm_dev")(h, vp(d_s), n, sp, op, None)); ts.append(time.perf_counter() - t0)
    DEPTH = 3
    def run(k):
        for i in range(k + DEPTH):
            if i >= DEPTH: zk.check(F("msm_collect")(h, (i - DEPTH) % 8, op, None))
            if i < k: zk.check(F("msm_submit")(h, vp(d_s), n, sp, i % 8))
    run(3); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); run(12); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 12)
    print("c %s %s 2^%d: single %.3f ms (median %.3f)  pipelined %.3f ms  checksum %x" % (os.environ.get("ZKT_MSM_C", "table"), grp, log2n, min(ts) * 1e3, sorted(ts)[4] * 1e3, best * 1e3, int(out.sum()) & 0xffffffff), flush=True)
    F("bases_free")(h)
