#!/usr/bin/env python3
"""Resident G1 MSM on small-valued scalars (a witness of bits, nibbles, bytes): every scalar of a window lands in one of 2^k buckets.  Single latency and pipelined time per MSM,
result checked against the python-integer sum.  usage: msm_skew.py <log2n> [<log2n> ...]"""
import ctypes, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
from zkt_testlib import G1_GEN, int_to_limbs, R, py_g1_mul, g1_arr
zk = importlib.import_module("zk-toolkit_amd"); zk.init(0); L = zk.lib()
dev = torch.device("cuda", 0); sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream); vp = lambda t: ctypes.c_void_p(t.data_ptr())
gen = np.zeros((1, 13), np.uint64); gen[0, :6] = int_to_limbs(G1_GEN[0], 6); gen[0, 6:12] = int_to_limbs(G1_GEN[1], 6)
for log2n in [int(x) for x in sys.argv[1:]]:
    n = 1 << log2n
    kb = bench.rand_scalars_mod_r(3, n)
    d_g = torch.from_numpy(np.repeat(gen, n, axis=0).view(np.int64)).to(dev); d_k = torch.from_numpy(kb.view(np.int64)).to(dev)
    d_b = torch.empty((n, 13), dtype=torch.int64, device=dev)
    zk.check(L.zkt_g1_mul_batch_dev(vp(d_g), vp(d_k), 4, vp(d_b), n, sp)); torch.cuda.synchronize()
    h = ctypes.c_void_p(); zk.check(L.zkt_g1_bases_from_device(vp(d_b), n, sp, ctypes.byref(h)))
    kint = [int(v) for v in (kb[:, 0].astype(object) + (kb[:, 1].astype(object) << 64) + (kb[:, 2].astype(object) << 128) + (kb[:, 3].astype(object) << 192))]
    for bits in (1, 4, 8, 255):
        if bits == 255: s = bench.rand_scalars_mod_r(4, n)
        else:
            s = np.zeros((n, 4), np.uint64); s[:, 0] = np.random.default_rng(5).integers(0, 1 << bits, n, dtype=np.uint64)
        d_s = torch.from_numpy(s.view(np.int64)).to(dev)
        out = np.zeros((1, 13), np.uint64); op = out.ctypes.data_as(ctypes.c_void_p)
        for _ in range(3): zk.check(L.zkt_g1_msm_dev(h, vp(d_s), n, sp, op, None))
        ts = []
        for _ in range(9):
            torch.cuda.synchronize(); t0 = time.perf_counter(); zk.check(L.zkt_g1_msm_dev(h, vp(d_s), n, sp, op, None)); ts.append(time.perf_counter() - t0)
        def run(k, D=3):
            for i in range(k + D):
                if i >= D: zk.check(L.zkt_g1_msm_collect(h, (i - D) % 8, op, None))
                if i < k: zk.check(L.zkt_g1_msm_submit(h, vp(d_s), n, sp, i % 8))
        run(3); torch.cuda.synchronize(); best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); run(12); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 12)
        ok = ""
        if bits != 255 and log2n <= 18:
            tot = sum(a * int(b) for a, b in zip(kint, s[:, 0])) % R
            ok = "  ok" if (out == g1_arr([py_g1_mul(G1_GEN, tot)])).all() else "  MISMATCH"
        print("g1 2^%d scalars < 2^%-3d: single %.3f ms (median %.3f)  pipelined %.3f ms%s" % (log2n, bits, min(ts) * 1e3, sorted(ts)[4] * 1e3, best * 1e3, ok), flush=True)
    L.zkt_g1_bases_free(h)
