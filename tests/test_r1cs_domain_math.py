"""The identity behind the scalable Groth16 path: on the reference's interpolation domain {1..n}
(qap.rs:33-97) the proof's Fr quantities a(x), b(x), h(x) t(x) can be computed from (A w, B w, C w) alone — no
coefficient-form polynomials.  Checked in python integers against the dense QAP of tests/qap_util.py."""
import pytest
from zkt_testlib import R, SplitMix64
from qap_util import example_cubic, chain_circuit, qap_from_r1cs, domain_model


def _eval(p, x):
    acc = 0
    for c in reversed(p): acc = (acc * x + c) % R
    return acc


@pytest.mark.parametrize("case", ["cubic", "chain1", "chain2", "chain7", "chain16"])
def test_domain_form_equals_dense_qap(case):
    A, B, C, w, l = example_cubic() if case == "cubic" else chain_circuit(int(case[5:]))
    ui, vi, wi, h, t = qap_from_r1cs(A, B, C, w)
    x = SplitMix64(99).below(R - 100) + 50
    comb = lambda P: sum(w[i] * _eval(P[i], x) for i in range(len(w))) % R
    ax, bx, htx = domain_model(A, B, C, w, x)
    assert ax == comb(ui) and bx == comb(vi)
    assert htx == _eval(h, x) * _eval(t, x) % R
