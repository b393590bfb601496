"""diagnostic: the inversion entry points one by one (run under rocgdb when one of them takes the process down)"""
import ctypes, importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from zkt_testlib import *
zk = importlib.import_module("zk-toolkit_amd"); zk.init(0); L = zk.lib()
rng = SplitMix64(5)
for deg, w in ((1, 6), (2, 12), (6, 36), (12, 72)):
    a = ints_to_arr([rng.below(Q) for _ in range(w // 6)], 6).reshape(1, w); o = np.zeros_like(a)
    name = "zkt_fq_inv_batch" if deg == 1 else f"zkt_fq{deg}_inv_batch"
    print(name, flush=True)
    print("  rc", getattr(L, name)(ptr(a), ptr(o), 1), hex(int(o.sum()) & 0xffffffff), flush=True)
