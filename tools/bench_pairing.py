#!/usr/bin/env python3
"""A/B helper: time zkt_tate_batch_dev for a batch on the GPU (inputs resident), using the library in ZKT_LIB_PATH."""
import ctypes, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from zkt_testlib import G1_GEN, G2_GEN, int_to_limbs
sys.path.insert(0, ROOT)
import bench
zk = importlib.import_module("zk-toolkit_amd"); zk.init(0); L = zk.lib()
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 14
dev = torch.device("cuda", 0); sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream); vp = lambda t: ctypes.c_void_p(t.data_ptr())
g1 = np.zeros((1, 13), np.uint64); g1[0, :6] = int_to_limbs(G1_GEN[0], 6); g1[0, 6:12] = int_to_limbs(G1_GEN[1], 6)
g2 = np.zeros((1, 25), np.uint64); (x1, x0), (y1, y0) = G2_GEN
g2[0, 0:6] = int_to_limbs(x1, 6); g2[0, 6:12] = int_to_limbs(x0, 6); g2[0, 12:18] = int_to_limbs(y1, 6); g2[0, 18:24] = int_to_limbs(y0, 6)
d_p = torch.empty((m, 13), dtype=torch.int64, device=dev); d_q = torch.empty((m, 25), dtype=torch.int64, device=dev)
keep = [torch.from_numpy(np.repeat(g1, m, axis=0).view(np.int64)).to(dev), torch.from_numpy(bench.rand_scalars_mod_r(5, m).view(np.int64)).to(dev),
        torch.from_numpy(np.repeat(g2, m, axis=0).view(np.int64)).to(dev), torch.from_numpy(bench.rand_scalars_mod_r(6, m).view(np.int64)).to(dev)]
zk.check(L.zkt_g1_mul_batch_dev(vp(keep[0]), vp(keep[1]), 4, vp(d_p), m, sp))
zk.check(L.zkt_g2_mul_batch_dev(vp(keep[2]), vp(keep[3]), 4, vp(d_q), m, sp))
torch.cuda.synchronize()
d_e = torch.empty((m, 72), dtype=torch.int64, device=dev)
if os.environ.get("ZKT_BENCH_PTRS"): print("ptrs g1 %x +%x  g2 %x +%x  out %x +%x" % (d_p.data_ptr(), d_p.numel() * 8, d_q.data_ptr(), d_q.numel() * 8, d_e.data_ptr(), d_e.numel() * 8), flush=True)
zk.check(L.zkt_tate_batch_dev(vp(d_p), vp(d_q), vp(d_e), m, sp)); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); zk.check(L.zkt_tate_batch_dev(vp(d_p), vp(d_q), vp(d_e), m, sp)); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("%s batch=%d best=%.2f ms -> %.0f pairings/s  checksum=%x" % (os.path.basename(zk.LIB_PATH), m, min(ts) * 1e3, m / min(ts), int(d_e.sum().item()) & 0xffffffffffff))
