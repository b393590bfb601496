"""diagnostic: the one-shot MSM entry points called many times on the same small inputs (with infinities and repeated bases): every call must give the
same bits as the first and as the oracle.  A difference means a race or an uninitialised read in the table-free path."""
import ctypes, importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from zkt_testlib import *
zk = importlib.import_module("zk-toolkit_amd"); zk.init(0); L = zk.lib(); O = oracle()
import torch
junk = torch.randint(-2**62, 2**62, (1 << 26,), dtype=torch.int64, device="cuda"); del junk       # leave garbage in freed device memory
bad = 0
for name, W, order, gen in (("g1", G1W, R, O.zkto_g1_generator), ("g2", G2W, R, O.zkto_g2_generator), ("secp", 9, SECP_N, O.zkto_secp_generator)):
    g = np.zeros((1, W), np.uint64); gen(ptr(g))
    for n in (1, 3, 4, 5, 17, 64, 300):
        rng = SplitMix64(1000 + n)
        ks = [rng.below(order) for _ in range(n)]
        if n >= 4: ks[1] = 0; ks[3] = ks[2]                      # an infinity and a repeated base
        bases = np.zeros((n, W), np.uint64)
        assert getattr(O, f"zkto_{name}_mul_batch")(ptr(np.repeat(g, n, axis=0)), ptr(ints_to_arr(ks, 4)), 4, ptr(bases), n, 8) == 0
        sc = ints_to_arr([rng.below(order) if i % 3 else rng.below(4) for i in range(n)], 4)
        tot = sum(k * limbs_to_int(s) for k, s in zip(ks, sc)) % order
        want = np.zeros((1, W), np.uint64)
        assert getattr(O, f"zkto_{name}_mul_batch")(ptr(g), ptr(ints_to_arr([tot], 4)), 4, ptr(want), 1, 1) == 0
        diff = 0
        for rep in range(150):
            got = np.zeros((1, W), np.uint64)
            zk.check(getattr(L, f"zkt_{name}_msm")(ptr(bases), ptr(sc), n, ptr(got)))
            if not (got == want).all(): diff += 1
        print(name, n, "mismatches", diff, flush=True); bad += diff
print("TOTAL mismatches", bad)
sys.exit(1 if bad else 0)
