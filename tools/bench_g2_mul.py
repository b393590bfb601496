#!/usr/bin/env python3
"""A/B helper: time the batched G1 / G2 scalar multiplications (zkt_g1_mul_batch_dev, zkt_g2_mul_batch_dev) on the GPU, inputs resident, using the library in ZKT_LIB_PATH."""
import ctypes, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from zkt_testlib import G1_GEN, G2_GEN, int_to_limbs
import bench
zk = importlib.import_module("zk-toolkit_amd"); zk.init(0); L = zk.lib()
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
dev = torch.device("cuda", 0); sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream); vp = lambda t: ctypes.c_void_p(t.data_ptr())
g1 = np.zeros((1, 13), np.uint64); g1[0, :6] = int_to_limbs(G1_GEN[0], 6); g1[0, 6:12] = int_to_limbs(G1_GEN[1], 6)
g2 = np.zeros((1, 25), np.uint64); (x1, x0), (y1, y0) = G2_GEN
g2[0, 0:6] = int_to_limbs(x1, 6); g2[0, 6:12] = int_to_limbs(x0, 6); g2[0, 12:18] = int_to_limbs(y1, 6); g2[0, 18:24] = int_to_limbs(y0, 6)
b1 = torch.from_numpy(np.repeat(g1, m, axis=0).view(np.int64)).to(dev); b2 = torch.from_numpy(np.repeat(g2, m, axis=0).view(np.int64)).to(dev)
k = torch.from_numpy(bench.rand_scalars_mod_r(5, m).view(np.int64)).to(dev)
o1 = torch.empty((m, 13), dtype=torch.int64, device=dev); o2 = torch.empty((m, 25), dtype=torch.int64, device=dev)
for name, fn, b, o in (("g1", L.zkt_g1_mul_batch_dev, b1, o1), ("g2", L.zkt_g2_mul_batch_dev, b2, o2)):
    zk.check(fn(vp(b), vp(k), 4, vp(o), m, sp)); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); zk.check(fn(vp(b), vp(k), 4, vp(o), m, sp)); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("%s %s_mul batch=%d best=%.2f ms -> %.2f M/s checksum=%x" % (os.path.basename(zk.LIB_PATH), name, m, min(ts) * 1e3, m / min(ts) / 1e6, int(o.sum().item()) & 0xffffffffffff))
