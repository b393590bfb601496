"""Oracle restatement of Pinocchio (pinocchio/crs.rs:49-161, prover.rs:98-170, verifier.rs:31-85) pinned the only way the
reference pins it: its own end-to-end test accepts (prover.rs:177-211, the cubic example, mid_beg = 3) — plus rejection of
tampered inputs, which the protocol demands.  CPU only."""
import ctypes
import numpy as np
from zkt_testlib import *
from qap_util import *

O = oracle()


def _run(A, B, C, wit, n_io, seed):
    n, n_mid = len(A), len(wit) - n_io
    V, W, Y, h, max_degree = pinocchio_instance(A, B, C, wit)
    rng = SplitMix64(seed)
    rnd = ints_to_arr([rng.below(R - 1) + 1 for _ in range(8)], 4)
    dv, dy = ints_to_arr([rng.below(R - 1) + 1], 4), ints_to_arr([rng.below(R - 1) + 1], 4)
    crs, cbuf = alloc_pinocchio(n, n_io, n_mid, max_degree)
    assert O.zkto_pinocchio_setup(ctypes.byref(crs), ptr(V), ptr(W), ptr(Y), ptr(rnd)) == 0
    pf, pbuf = alloc_pinocchio_proof()
    wires, H = ints_to_arr(wit, 4), ints_to_arr(h, 4)
    assert O.zkto_pinocchio_prove(ctypes.byref(crs), ptr(wires), ptr(H), len(h), ptr(dv), ptr(dy), ctypes.byref(pf)) == 0
    return crs, cbuf, pf, pbuf, wires


def test_reference_example_accepts_and_tampering_rejects():
    A, B, C, wit, l = example_cubic()
    crs, cbuf, pf, pbuf, wires = _run(A, B, C, wit, l + 1, 31)                    # witness.io() = wires 0..mid_beg-1 (witness.rs:20-23)
    io = wires[:l + 1].copy()
    assert O.zkto_pinocchio_verify(ctypes.byref(crs), ctypes.byref(pf), ptr(io)) == 1
    bad = io.copy(); bad[2, 0] ^= np.uint64(1)                                   # a different public output
    assert O.zkto_pinocchio_verify(ctypes.byref(crs), ctypes.byref(pf), ptr(bad)) == 0
    keep = pbuf["alpha_w_mid_s"].copy()
    pbuf["alpha_w_mid_s"][:] = pbuf["alpha_v_mid_s"]                             # breaks the knowledge-of-coefficient check of w
    assert O.zkto_pinocchio_verify(ctypes.byref(crs), ctypes.byref(pf), ptr(io)) == 0
    pbuf["alpha_w_mid_s"][:] = keep
    assert O.zkto_pinocchio_verify(ctypes.byref(crs), ctypes.byref(pf), ptr(io)) == 1


def test_chain_circuit_accepts():
    A, B, C, wit, l = chain_circuit(4)
    crs, cbuf, pf, pbuf, wires = _run(A, B, C, wit, l + 1, 32)
    assert O.zkto_pinocchio_verify(ctypes.byref(crs), ctypes.byref(pf), ptr(wires[:l + 1].copy())) == 1
