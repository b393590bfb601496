#!/usr/bin/env python3
"""A/B helper: pipelined G2 MSM over resident bases (n = 2^log2n), library from ZKT_LIB_PATH.  Prints ms per MSM and a checksum."""
import ctypes, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from zkt_testlib import G2_GEN, int_to_limbs
import bench
zk = importlib.import_module("zk-toolkit_amd"); zk.init(0); L = zk.lib()
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n = 1 << log2n
dev = torch.device("cuda", 0); sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream); vp = lambda t: ctypes.c_void_p(t.data_ptr())
g2 = np.zeros((1, 25), np.uint64); (x1, x0), (y1, y0) = G2_GEN
g2[0, 0:6] = int_to_limbs(x1, 6); g2[0, 6:12] = int_to_limbs(x0, 6); g2[0, 12:18] = int_to_limbs(y1, 6); g2[0, 18:24] = int_to_limbs(y0, 6)
d_g = torch.from_numpy(np.repeat(g2, n, axis=0).view(np.int64)).to(dev)
d_k = torch.from_numpy(bench.rand_scalars_mod_r(3, n).view(np.int64)).to(dev)
d_b = torch.empty((n, 25), dtype=torch.int64, device=dev)
zk.check(L.zkt_g2_mul_batch_dev(vp(d_g), vp(d_k), 4, vp(d_b), n, sp)); torch.cuda.synchronize()
h = ctypes.c_void_p(); zk.check(L.zkt_g2_bases_from_device(vp(d_b), n, sp, ctypes.byref(h)))
d_s = torch.from_numpy(bench.rand_scalars_mod_r(4, n).view(np.int64)).to(dev)
out = np.zeros((1, 25), np.uint64); op = out.ctypes.data_as(ctypes.c_void_p)
DEPTH = 3
def run(k):
    for i in range(k + DEPTH):
        if i >= DEPTH: zk.check(L.zkt_g2_msm_collect(h, (i - DEPTH) % 8, op, None))
        if i < k: zk.check(L.zkt_g2_msm_submit(h, vp(d_s), n, sp, i % 8))
run(3); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / steps)
print("%s G2 MSM n=2^%d: %.2f ms per MSM (pipelined, best of 3)  checksum=%x" % (os.path.basename(zk.LIB_PATH), log2n, best * 1e3, int(out.sum()) & 0xffffffffffff))
L.zkt_g2_bases_free(h)
