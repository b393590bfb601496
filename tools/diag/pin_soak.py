"""Soak run for the one unexplained bit mismatch of round 3 (gpurun_out/r03i_gputests.log: test_pinocchio_vs_oracle[cubic], proof element g2_w_mid_s, in the child
process with the large-batch kernels forced and graphs off): the very call sequence of that test — zkt_pinocchio_setup once, then zkt_pinocchio_prove N times on the
same inputs, every proof element compared with the oracle's — plus the one-shot G2 MSM behind g2_w_mid_s on its own, N times.  Run with
  ZKT_DTATE_MAX=0 ZKT_DPRODUCT_MAX=0 ZKT_MSM_GRAPH=0 ZKT_DEBUG_POISON=1 python3 tools/diag/pin_soak.py [N]
Prints one line per mismatch (element, repetition, first differing word, whether the next repetition agrees) and a total."""
import ctypes, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from zkt_testlib import *
from qap_util import *
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
zk = importlib.import_module("zk-toolkit_amd"); zk.init(0); L = zk.lib(); O = oracle()
import torch
bad = 0
for case in ("cubic", "chain4", "chain9"):
    A, B, C, wit, l = example_cubic() if case == "cubic" else chain_circuit(int(case[5:]))
    n, n_io = len(A), l + 1
    n_mid = len(wit) - n_io
    V, W, Y, h, max_degree = pinocchio_instance(A, B, C, wit)
    rng = SplitMix64(77 + n)
    rnd = ints_to_arr([rng.below(R - 1) + 1 for _ in range(8)], 4)
    dv, dy = ints_to_arr([rng.below(R - 1) + 1], 4), ints_to_arr([rng.below(R - 1) + 1], 4)
    wires, H = ints_to_arr(wit, 4), ints_to_arr(h, 4)
    ocrs, obuf = alloc_pinocchio(n, n_io, n_mid, max_degree); gcrs, gbuf = alloc_pinocchio(n, n_io, n_mid, max_degree)
    assert O.zkto_pinocchio_setup(ctypes.byref(ocrs), ptr(V), ptr(W), ptr(Y), ptr(rnd)) == 0
    zk.check(L.zkt_pinocchio_setup(ctypes.byref(gcrs), ptr(V), ptr(W), ptr(Y), ptr(rnd)))
    for k in obuf: assert (obuf[k] == gbuf[k]).all(), f"CRS field {k} differs"
    opf, opb = alloc_pinocchio_proof()
    assert O.zkto_pinocchio_prove(ctypes.byref(ocrs), ptr(wires), ptr(H), len(h), ptr(dv), ptr(dy), ctypes.byref(opf)) == 0
    last_bad = {}
    for rep in range(N):
        if rep % 16 == 5:                                  # churn the allocator between proofs, as a suite does
            junk = torch.randint(-2**62, 2**62, (1 << 22,), dtype=torch.int64, device="cuda"); del junk; torch.cuda.empty_cache()
        gpf, gpb = alloc_pinocchio_proof()
        zk.check(L.zkt_pinocchio_prove(ctypes.byref(gcrs), ptr(wires), ptr(H), len(h), ptr(dv), ptr(dy), ctypes.byref(gpf)))
        for k in opb:
            if not (opb[k] == gpb[k]).all():
                bad += 1
                print(f"MISMATCH {case} rep {rep} element {k} first differing word {int(np.argmax((opb[k] != gpb[k]).ravel()))} of {opb[k].size} (previous mismatch of this element at rep {last_bad.get(k)})", flush=True)
                last_bad[k] = rep
    # the sum behind g2_w_mid_s on its own: sum over the mid wires of wire * g2_wk_mid (prover.rs:121-141), one-shot G2 MSM with host pointers
    if n_mid:
        want = np.zeros((1, G2W), np.uint64); zk.check(L.zkt_g2_msm(ptr(gbuf["g2_wk_mid"]), ptr(wires[n_io:].copy()), n_mid, ptr(want)))
        for rep in range(2 * N):
            got = np.zeros((1, G2W), np.uint64); zk.check(L.zkt_g2_msm(ptr(gbuf["g2_wk_mid"]), ptr(wires[n_io:].copy()), n_mid, ptr(got)))
            if not (got == want).all(): bad += 1; print(f"MISMATCH {case} one-shot G2 MSM rep {rep}", flush=True)
    print(f"{case}: {N} proofs, {2 * N if n_mid else 0} one-shot G2 sums done, mismatches so far {bad}", flush=True)
print("TOTAL mismatches", bad)
sys.exit(1 if bad else 0)
