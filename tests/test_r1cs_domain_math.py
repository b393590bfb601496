"""The identity behind the scalable Groth16 path: on the reference's interpolation domain {1..n}
(qap.rs:33-97) the proof's Fr quantities a(x), b(x), h(x) t(x) can be computed from (A w, B w, C w) alone — no
coefficient-form polynomials.  Checked in python integers against the dense QAP of tests/qap_util.py."""
import pytest
from zkt_testlib import R, SplitMix64
import numpy as np
from zkt_testlib import ints_to_arr
from qap_util import example_cubic, chain_circuit, chain_circuit_sparse, qap_from_r1cs, domain_model, groth16_proof_scalars


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


@pytest.mark.parametrize("n", [1, 2, 12, 40])
def test_proof_scalars_in_linear_time_equal_the_coefficient_form(n):
    """groth16_proof_scalars (the O(n) python-integer checker of the full-size GPU tests: discrete logarithms of A, B, C from the trapdoor) against the
    reference's own coefficient-form prover (prover.rs:96-147 over the dense QAP of qap_from_r1cs; crs.rs:65-121 for uvw_wit and xt_by_delta)."""
    mats, wires, l, m = chain_circuit_sparse(n, seed=11)
    A, B, C, wit, l2 = chain_circuit(n, seed=11)
    toi = lambda a: [int.from_bytes(np.ascontiguousarray(v).tobytes(), "little") for v in np.asarray(a).reshape(-1, 4)]
    assert toi(wires) == [v % R for v in wit] and l == l2
    rng = SplitMix64(5 + n); fr = lambda v: ints_to_arr([v], 4)
    trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]; r = fr(rng.below(R - 1) + 1); s = fr(rng.below(R - 1) + 1)
    got = groth16_proof_scalars(mats, wires, l, trap, r, s)
    alpha, beta, gamma, delta, x = [toi(t)[0] for t in trap]
    rr, ss = toi(r)[0], toi(s)[0]
    ui, vi, wi, h, t = qap_from_r1cs(A, B, C, wit)
    As = (alpha + sum(wit[i] * _eval(ui[i], x) for i in range(m + 1)) + rr * delta) % R
    Bs = (beta + sum(wit[i] * _eval(vi[i], x) for i in range(m + 1)) + ss * delta) % R
    di = pow(delta, -1, R)
    Cs = (sum(wit[i] * (beta * _eval(ui[i], x) + alpha * _eval(vi[i], x) + _eval(wi[i], x)) for i in range(l + 1, m + 1)) * di
          + _eval(h, x) * _eval(t, x) * di + ss * As + rr * Bs - rr * ss * delta) % R
    assert got == (As, Bs, Cs)
