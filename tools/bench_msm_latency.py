import ctypes, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
from zkt_testlib import G1_GEN, int_to_limbs
zk = importlib.import_module("zk-toolkit_amd"); zk.init(0); L = zk.lib()
dev = torch.device("cuda", 0); sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream); vp = lambda t: ctypes.c_void_p(t.data_ptr())
for log2n in (20, 17):
    n = 1 << log2n
    gen = np.zeros((1, 13), dtype=np.uint64); gen[0, :6] = int_to_limbs(G1_GEN[0], 6); gen[0, 6:12] = int_to_limbs(G1_GEN[1], 6)
    d_gen = torch.from_numpy(np.repeat(gen, n, axis=0).view(np.int64)).to(dev)
    d_k = torch.from_numpy(bench.rand_scalars_mod_r(3, n).view(np.int64)).to(dev)
    d_b = torch.empty((n, 13), dtype=torch.int64, device=dev)
    zk.check(L.zkt_g1_mul_batch_dev(vp(d_gen), vp(d_k), 4, vp(d_b), n, sp)); torch.cuda.synchronize()
    h = ctypes.c_void_p(); zk.check(L.zkt_g1_bases_from_device(vp(d_b), n, sp, ctypes.byref(h)))
    d_s = torch.from_numpy(bench.rand_scalars_mod_r(4, n).view(np.int64)).to(dev)
    out = np.zeros((1, 13), np.uint64); op = out.ctypes.data_as(ctypes.c_void_p)
    for _ in range(3): zk.check(L.zkt_g1_msm_dev(h, vp(d_s), n, sp, op, None))
    ts = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); zk.check(L.zkt_g1_msm_dev(h, vp(d_s), n, sp, op, None)); ts.append(time.perf_counter() - t0)
    print("G1 single MSM latency 2^%d: best %.3f ms median %.3f ms  acc %.3f ms  checksum %x" % (log2n, min(ts) * 1e3, sorted(ts)[5] * 1e3, L.zkt_last_kernel_ms(), int(out.sum()) & 0xffffffff))
    L.zkt_g1_bases_free(h)
