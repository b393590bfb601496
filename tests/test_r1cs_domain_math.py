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


def _block_plan(n, nshards, shard):
    """The host arithmetic of zkt_groth16_setup_r1cs_sharded (csrc/zkt_groth16_r1cs.hip): this rank's range of the n - 1 quotient values and its block sizes."""
    tot = n - 1 if n >= 2 else 0
    base, extra = tot // nshards, tot % nshards
    lo = shard * base + min(shard, extra); hi = lo + base + (1 if shard < extra else 0)
    cnt, s0 = hi - lo, lo + 1
    logM = 1
    while (1 << (logM - 1)) < cnt or (logM <= 10 and (1 << (logM - 1)) < n): logM += 1
    M = 1 << logM; Bi = M // 2; Q = (n + Bi - 1) // Bi if cnt else 1
    return cnt, s0, Bi, M, Q


@pytest.mark.parametrize("n,nshards", [(2, 1), (5, 1), (37, 1), (1025, 1), (1500, 1), (1500, 3), (2600, 5), (2600, 8), (40, 39)])
def test_blocked_convolution_gives_every_rank_its_quotient_values(n, nshards):
    """k_recip_blocks / k_prep_blocks / k_sum_blocks in python integers: S(s) = sum_{j=1..n} f_j / (n + s - j) for the rank's s = s0 .. s0 + cnt - 1 equals entry s - s0 of
    sum_q (block q of f, zero-padded to M) (*) (kernel slice q) as CYCLIC convolutions of size M = 2 Bi — the overlap-save identity the sharded quotient stage rests on
    (prover.rs:64-71 divides in coefficient form; qap.rs:33-97 fixes the domain {1..n}).  The cyclic convolutions are evaluated directly (no transform), sampled outputs."""
    rng = SplitMix64(1000 * n + nshards)
    f = [0] + [rng.below(R) for _ in range(n)]                        # f[j], j = 1..n
    inv = lambda d: pow(d, -1, R)
    g = lambda d: inv(d) if 1 <= d <= 2 * n - 1 else 0
    seen = 0
    for shard in range(nshards):
        cnt, s0, Bi, M, Q = _block_plan(n, nshards, shard)
        assert cnt <= Bi and Q * Bi >= n
        if cnt == 0: continue
        # kernel slices exactly as k_recip_blocks lays them out
        K = []
        for q in range(Q):
            base = n + s0 - 1 - q * Bi
            K.append({e: g(base + e) for e in range(cnt)} | {e: g(base - (M - e)) for e in range(M - Bi + 1, M)})
        X = [[f[q * Bi + 1 + e] if e < Bi and q * Bi + e < n else 0 for e in range(M)] for q in range(Q)]
        for u in sorted({0, cnt - 1, cnt // 2, rng.below(cnt)}):     # sampled outputs: a cyclic convolution entry costs M products per block
            got = 0
            for q in range(Q):
                for v in range(Bi):                                   # the upper half of a block is zero padding
                    if X[q][v]: got += X[q][v] * K[q].get((u - v) % M, 0)
            s = s0 + u
            want = sum(f[j] * inv(n + s - j) for j in range(1, n + 1)) % R
            assert got % R == want, (shard, u)
        seen += cnt
    assert seen == max(n - 1, 0)                                      # the ranks' ranges tile s = 1 .. n-1
