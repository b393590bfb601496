"""The lane-distributed small-batch pairing (csrc/zkt_dpairing.hip): its Fq12 building blocks against zkt_fq12_*_batch / the oracle, then
the pairing itself against zkt_tate_batch's large-batch kernel and the oracle at the batch sizes that select it."""
import ctypes, importlib
import numpy as np
import pytest
from zkt_testlib import *

pytestmark = pytest.mark.gpu
zk = importlib.import_module("zk-toolkit_amd")
O = oracle()


@pytest.fixture(scope="module")
def L():
    zk.init()
    return zk.lib()


def _rand_fq12(seed, n):
    rng = SplitMix64(seed)
    return ints_to_arr([rng.below(Q) for _ in range(12 * n)], 6).reshape(n, FQ12)


@pytest.mark.parametrize("n", [1, 5, 6, 23])
def test_distributed_fq12_ops(L, n):
    a, b = _rand_fq12(1, n), _rand_fq12(2, n)
    a[0, :] = 0; a[0, 66:72] = ints_to_arr([3], 6)[0]                       # the scalar 3 embedded in w0.v0.u0 (fq12.rs:60-67)
    got, want = np.zeros_like(a), np.zeros_like(a)
    zk.check(L.zkt_debug_dfq12_op(0, ptr(a), ptr(b), ptr(got), n)); zk.check(L.zkt_fq12_mul_batch(ptr(a), ptr(b), ptr(want), n))
    assert (got == want).all(), "mul"
    assert O.zkto_fq12_op(2, ptr(a), ptr(b), ptr(want), n) == 0 and (got == want).all()
    zk.check(L.zkt_debug_dfq12_op(1, ptr(a), None, ptr(got), n)); zk.check(L.zkt_fq12_mul_batch(ptr(a), ptr(a), ptr(want), n))
    assert (got == want).all(), "square"
    zk.check(L.zkt_debug_dfq12_op(5, ptr(b), None, ptr(got), n)); zk.check(L.zkt_fq12_inv_batch(ptr(b), ptr(want), n))
    assert (got == want).all(), "inverse"
    for op, k in ((2, 1), (3, 2), (4, 6)):                                    # a^(q^k) through Fq12::pow (fq12.rs:42-57) on the library's large-batch path
        e = Q ** k
        limbs = np.array([(e >> (32 * i)) & 0xFFFFFFFF for i in range((e.bit_length() + 31) // 32)], dtype=np.uint32)
        zk.check(L.zkt_debug_dfq12_op(op, ptr(a), None, ptr(got), n))
        zk.check(L.zkt_fq12_pow_batch(ptr(a), limbs.ctypes.data_as(ctypes.c_void_p), len(limbs), ptr(want), n))
        assert (got == want).all(), f"frobenius^{k}"
    # Granger-Scott squaring on elements of the cyclotomic subgroup: a^((q^6-1)(q^2+1)), then square both ways
    e = (Q ** 6 - 1) * (Q ** 2 + 1)
    limbs = np.array([(e >> (32 * i)) & 0xFFFFFFFF for i in range((e.bit_length() + 31) // 32)], dtype=np.uint32)
    cyc = np.zeros_like(a)
    zk.check(L.zkt_fq12_pow_batch(ptr(b), limbs.ctypes.data_as(ctypes.c_void_p), len(limbs), ptr(cyc), n))
    zk.check(L.zkt_debug_dfq12_op(6, ptr(cyc), None, ptr(got), n)); zk.check(L.zkt_fq12_mul_batch(ptr(cyc), ptr(cyc), ptr(want), n))
    assert (got == want).all(), "cyclotomic square"


@pytest.mark.parametrize("n", [1, 4, 5, 11, 64])
def test_small_batch_tate_vs_oracle(L, n):
    """zkt_tate_batch at batch sizes that take the lane-distributed kernel: bit-identical to the oracle's reference algorithm (pairing.rs:86-100)"""
    rng = SplitMix64(900 + n)
    g1 = np.zeros((1, G1W), np.uint64); O.zkto_g1_generator(ptr(g1))
    g2 = np.zeros((1, G2W), np.uint64); O.zkto_g2_generator(ptr(g2))
    P = np.zeros((n, G1W), np.uint64); Qs = np.zeros((n, G2W), np.uint64)
    zk.check(L.zkt_g1_mul_batch(ptr(np.repeat(g1, n, axis=0)), ptr(ints_to_arr([rng.below(R - 1) + 1 for _ in range(n)], 4)), 4, ptr(P), n))
    zk.check(L.zkt_g2_mul_batch(ptr(np.repeat(g2, n, axis=0)), ptr(ints_to_arr([rng.below(R - 1) + 1 for _ in range(n)], 4)), 4, ptr(Qs), n))
    if n >= 4: P[2] = g1[0]; Qs[2] = g2[0]                                   # tate(G1, G2): SURVEY Appendix B value
    got, want = np.zeros((n, FQ12), np.uint64), np.zeros((n, FQ12), np.uint64)
    zk.check(L.zkt_tate_batch(ptr(P), ptr(Qs), ptr(got), n))
    m = min(n, 16)
    assert O.zkto_pairing_batch(3, ptr(P[:m].copy()), ptr(Qs[:m].copy()), ptr(want[:m]), m, 16, None) == 0
    assert (got[:m] == want[:m]).all()
    if n > m:                                                                # the rest by bilinearity against the first element is overkill: compare with a second call in another order
        perm = np.arange(n)[::-1].copy()
        got2 = np.zeros_like(got)
        zk.check(L.zkt_tate_batch(ptr(P[perm].copy()), ptr(Qs[perm].copy()), ptr(got2), n))
        assert (got2[perm.argsort()] == got).all()
    # infinity: the reference's panic (rational_function.rs:36,59), with the element's index
    if n >= 4:
        P2 = P.copy(); P2[3] = 0; P2[3, 12] = 1
        assert L.zkt_tate_batch(ptr(P2), ptr(Qs), ptr(got), n) == ZKT_ERR_INFINITY and L.zkt_last_error_index() == 3


def test_large_batch_kernels_still_covered_in_a_forced_process():
    """Small batches now take the lane-distributed kernels, so the one-element-per-lane kernels (k_tate, k_groth16_verify_ate with the key's line and
    statement tables — the chain circuit's statement is a full-size field element —, k_pairing_product_check_ate and the older kernels behind them)
    would only be reached by the 2^16-element tests.  Re-run the small parity tests of every pairing consumer in ONE
    child process with the switch-over forced to zero (ZKT_DTATE_MAX = ZKT_DPRODUCT_MAX = 0): same oracle, other kernels."""
    import os, subprocess, sys
    env = dict(os.environ, ZKT_DTATE_MAX="0", ZKT_DPRODUCT_MAX="0", ZKT_MSM_GRAPH="0")      # ... and the small MSMs of these protocols issued launch by launch, not as graph replays
    here = os.path.dirname(os.path.abspath(__file__))
    sel = "test_small_batch_tate_vs_oracle or verify_batch_mixed or groth16_chain_circuit or outside_the_subgroup or signature or sign_verify or pinocchio_vs_oracle and cubic or tate_and_weil or bilinear"
    r = subprocess.run([sys.executable, "-m", "pytest", here, "-m", "gpu", "-x", "-q", "-k", sel, "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_127_step_verification_kernels_still_covered_in_a_forced_process():
    """The deciding entry points run the 63-step loop by default (lane-distributed kernels for small batches, one element per lane for large ones); ZKT_PRODUCT_LOOP=127 keeps
    the 127-step kernels of round 2 for A/B measurements.  Re-run the signature and product-check parity tests on them in one child process: same oracle, older kernels."""
    import os, subprocess, sys
    env = dict(os.environ, ZKT_PRODUCT_LOOP="127")
    here = os.path.dirname(os.path.abspath(__file__))
    sel = "verify_batch_matches_reference_decision or pairing_product_check or pinocchio_vs_oracle and cubic"
    r = subprocess.run([sys.executable, "-m", "pytest", here, "-m", "gpu", "-x", "-q", "-k", sel, "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
