"""The affine pair-tree rounds ahead of the G2 bucket accumulation (zk-toolkit_amd/csrc/zkt_msm_affine.hip; Polynomial::eval_with_g2_hidings,
polynomial.rs:283-293, summed with the reference's own affine addition macros.rs:34-163 and one shared inversion per lane).  Measured inside the product they lose
to the lane-pair XYZZ kernel (profiles/r04_batched_affine_go_no_go.md), so they are OFF by default and kept as a tested alternative: the resident-base tests below
run in this process on the default plan, and once more in a child process with the rounds FORCED for every resident G2 sum (ZKT_G2_AFFINE_ROUNDS=2,
ZKT_G2_AFFINE_MIN_ENTRIES=1) — same oracle, every exceptional case of the addition (repeated bases: the tangent; a base and its negative: infinity; a base at
infinity) inside the rounds.  (All of 1, 2 and 3 rounds passed when the path was brought up; one child keeps the suite short.)"""
import ctypes, importlib, os, subprocess, sys
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


def _oracle_sum(bases, sc):
    n = len(bases)
    tmp = np.zeros_like(bases); assert O.zkto_g2_mul_batch(ptr(bases), ptr(sc), 4, ptr(tmp), n, 8) == 0
    acc = np.zeros((1, G2W), np.uint64); acc[0, G2W - 1] = 1
    for i in range(n):
        nxt = np.zeros_like(acc); assert O.zkto_g2_add_batch(ptr(acc), ptr(tmp[i:i + 1].copy()), ptr(nxt), 1) == 0; acc = nxt
    return acc


def _resident_msm(L, bases, sc):
    import torch
    n = len(bases)
    h = ctypes.c_void_p(); zk.check(L.zkt_g2_bases_upload(ptr(bases), n, ctypes.byref(h)))
    try:
        d = torch.from_numpy(sc.view(np.int64)).cuda()
        outs = []
        for _ in range(2):                                  # twice: the second call of a small set replays the captured graph
            got = np.zeros((1, G2W), np.uint64)
            zk.check(L.zkt_g2_msm_dev(h, ctypes.c_void_p(d.data_ptr()), n, None, ptr(got), None))
            outs.append(got)
        assert (outs[0] == outs[1]).all()
        return outs[0]
    finally:
        L.zkt_g2_bases_free(h)


@pytest.mark.parametrize("n", [1, 2, 3, 17, 100, 257])
def test_resident_g2_msm_vs_oracle(L, n):
    rng = SplitMix64(7700 + n)
    g = np.zeros((1, G2W), np.uint64); O.zkto_g2_generator(ptr(g))
    bases = np.zeros((n, G2W), np.uint64)
    zk.check(L.zkt_g2_mul_batch(ptr(np.repeat(g, n, axis=0)), ptr(ints_to_arr([rng.below(R - 1) + 1 for _ in range(n)], 4)), 4, ptr(bases), n))
    ss = [rng.below(R) for _ in range(n)]
    if n > 2: ss[1] = 0
    if n > 16: ss[5] = 1; ss[6] = R - 1; ss[7] = (1 << 256) - 1
    sc = ints_to_arr(ss, 4)
    assert (_resident_msm(L, bases, sc) == _oracle_sum(bases, sc)).all()


@pytest.mark.parametrize("kind", ["pool", "ones", "bits", "same-point"])
def test_resident_g2_msm_exceptional_cases(L, kind):
    """Equal points in one bucket (P + P: the tangent, macros.rs:57-108), opposite points (P + (-P) = infinity, macros.rs:53-56), bases at infinity, and buckets that hold
    almost everything (all scalars one / 0-1: the carry-free digit puts every term of a window into one bucket) — inside the affine rounds when they are forced."""
    n = 300
    rng = SplitMix64(8800)
    g = np.zeros((1, G2W), np.uint64); O.zkto_g2_generator(ptr(g))
    if kind in ("pool", "same-point"):
        pool = np.zeros((6, G2W), np.uint64)
        assert O.zkto_g2_mul_batch(ptr(np.repeat(g, 3, axis=0)), ptr(ints_to_arr([rng.below(R - 1) + 1 for _ in range(3)], 4)), 4, ptr(pool[:3]), 3, 3) == 0
        assert O.zkto_g2_mul_batch(ptr(pool[:2].copy()), ptr(ints_to_arr([R - 1] * 2, 4)), 4, ptr(pool[3:5]), 2, 2) == 0        # -P = (r - 1) P
        pool[5] = 0; pool[5, G2W - 1] = 1
        if kind == "pool":
            bases = np.stack([pool[rng.below(6)] for _ in range(n)])
            choices = [0, 1, 2, R - 1, 1, 2] + [rng.below(R) for _ in range(3)]
            ss = [choices[rng.below(len(choices))] for _ in range(n)]
        else:
            bases = np.repeat(pool[:1], n, axis=0)            # ONE point n times, one scalar: every pair of every round is a doubling
            ss = [12345] * n
    else:
        bases = np.zeros((n, G2W), np.uint64)
        zk.check(L.zkt_g2_mul_batch(ptr(np.repeat(g, n, axis=0)), ptr(ints_to_arr([rng.below(R - 1) + 1 for _ in range(n)], 4)), 4, ptr(bases), n))
        ss = [1] * n if kind == "ones" else [rng.below(2) for _ in range(n)]
    sc = ints_to_arr(ss, 4)
    assert (_resident_msm(L, bases, sc) == _oracle_sum(bases, sc)).all()


@pytest.mark.parametrize("rounds", [2])
def test_affine_rounds_forced_in_a_child_process(rounds):
    """Every resident G2 sum of the child takes `rounds` affine rounds: the tests of this file, the eight-slot pipeline, the sharded partials and the G2 leg of the
    Groth16 / Pinocchio provers (whose B sums are resident G2 sets) against the same oracle."""
    if os.environ.get("ZKT_G2_AFFINE_MIN_ENTRIES"): pytest.skip("already inside the forced child")
    env = dict(os.environ, ZKT_G2_AFFINE_MIN_ENTRIES="1", ZKT_G2_AFFINE_ROUNDS=str(rounds))
    here = os.path.dirname(os.path.abspath(__file__))
    sel = ("resident_g2_msm or (msm_eight_slots_in_flight and g2) or (sharded_msm_partials_combine and g2) or (r1cs_path_matches_reference_algorithm and (chain16 or bits61 or cubic))"
           " or pinocchio_resident_prover")
    r = subprocess.run([sys.executable, "-m", "pytest", here, "-m", "gpu", "-x", "-q", "-k", sel, "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
