"""Test-side QAP construction (python ints, Fr): R1CS -> QAP by Lagrange interpolation on the domain {1..n}
(the reference's choice, qap/qap.rs:33-97), t = prod (x - i) (qap.rs:115-135), h = (A*B - C)/t (prover.rs:64-71).
The reference builds these from an equation parser (out of scope); the hot path only consumes
the resulting coefficient arrays."""
import ctypes
import numpy as np
from zkt_testlib import R, ints_to_arr, G1W, G2W, FQ12, ptr


def poly_mul(a, b):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % R
    return out


def lagrange_basis(n):
    """coefficients (low first) of L_j(x), j=1..n, over points 1..n"""
    full = [1]
    for i in range(1, n + 1):
        full = poly_mul(full, [(-i) % R, 1])          # t(x)
    basis = []
    for j in range(1, n + 1):
        # synthetic division of t by (x - j)
        q = [0] * n; carry = 0
        for k in range(n, 0, -1):
            carry = (full[k] + carry * j) % R; q[k - 1] = carry
        denom = 1
        for i in range(1, n + 1):
            if i != j: denom = denom * (j - i) % R
        inv = pow(denom, -1, R)
        basis.append([c * inv % R for c in q])
    return basis, full


def qap_from_r1cs(Amat, Bmat, Cmat, witness):
    """A,B,C: n x (m+1) integer matrices; returns ui,vi,wi ((m+1) x n coefficient lists), h (list), t (list)"""
    n, cols = len(Amat), len(Amat[0])
    basis, t = lagrange_basis(n)
    def interp(M):
        polys = []
        for i in range(cols):
            p = [0] * n
            for j in range(n):
                v = M[j][i] % R
                if v:
                    for k in range(n): p[k] = (p[k] + v * basis[j][k]) % R
            polys.append(p)
        return polys
    ui, vi, wi = interp(Amat), interp(Bmat), interp(Cmat)
    comb = lambda P: [sum(witness[i] * P[i][k] for i in range(cols)) % R for k in range(n)]
    a, b, c = comb(ui), comb(vi), comb(wi)
    p = poly_mul(a, b)
    for k in range(n): p[k] = (p[k] - c[k]) % R
    # divide p by t (monic, degree n)
    p = p[:]; h = [0] * (len(p) - n)
    for k in range(len(p) - 1, n - 1, -1):
        q = p[k]; h[k - n] = q
        if q:
            for d in range(n + 1): p[k - n + d] = (p[k - n + d] - q * t[d]) % R
    assert all(v == 0 for v in p[:n]), "R1CS not satisfied"
    while len(h) > 1 and h[-1] == 0: h.pop()
    return ui, vi, wi, h, t


def example_cubic():
    """(x*x*x) + x + 5 == 35 with x = 3 — the reference's own Groth16 test (prover.rs:159-192): 7 wires, l = 2."""
    w = [1, 3, 35, 9, 27, 8, 35]           # one, x, out | t1=x*x, t2=t1*x, t3=x+5, t4=t2+t3   (wires.rs:47-54)
    A = [[0, 1, 0, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0, 0], [5, 1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 1, 1, 0], [0, 0, 0, 0, 0, 0, 1]]
    B = [[0, 1, 0, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0]]
    C = [[0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 1, 0], [0, 0, 0, 0, 0, 0, 1], [0, 0, 1, 0, 0, 0, 0]]
    return A, B, C, w, 2


def chain_circuit(n, seed=7):
    """n constraints w_{j+1} = w_j*w_j + c_j; wires [one, out | w_0 .. w_{n-1}], l = 1."""
    from zkt_testlib import SplitMix64
    rng = SplitMix64(seed)
    cs = [rng.below(1 << 32) for _ in range(n)]
    ws = [rng.below(R)]
    for j in range(n): ws.append((ws[j] * ws[j] + cs[j]) % R)
    wit = [1, ws[n]] + ws[:n]
    cols = n + 2
    idx = lambda j: 2 + j if j < n else 1
    A = [[0] * cols for _ in range(n)]; B = [[0] * cols for _ in range(n)]; C = [[0] * cols for _ in range(n)]
    for j in range(n):
        A[j][idx(j)] = 1; B[j][idx(j)] = 1; C[j][idx(j + 1)] = 1; C[j][0] = (-cs[j]) % R
    return A, B, C, wit, 1


class Crs(ctypes.Structure):
    _fields_ = [("n", ctypes.c_size_t), ("l", ctypes.c_size_t), ("m", ctypes.c_size_t)] + \
               [(k, ctypes.POINTER(ctypes.c_uint64)) for k in ("g1_alpha", "g1_beta", "g1_delta", "g1_xi", "g1_uvw_stmt", "g1_uvw_wit", "g1_xt_by_delta",
                                                                "g2_beta", "g2_gamma", "g2_delta", "g2_xi", "gt_alpha_beta")]


def alloc_crs(n, l, m):
    bufs = {"g1_alpha": np.zeros((1, G1W), np.uint64), "g1_beta": np.zeros((1, G1W), np.uint64), "g1_delta": np.zeros((1, G1W), np.uint64),
            "g1_xi": np.zeros((n, G1W), np.uint64), "g1_uvw_stmt": np.zeros((l + 1, G1W), np.uint64), "g1_uvw_wit": np.zeros((max(m - l, 1), G1W), np.uint64),
            "g1_xt_by_delta": np.zeros((n, G1W), np.uint64), "g2_beta": np.zeros((1, G2W), np.uint64), "g2_gamma": np.zeros((1, G2W), np.uint64),
            "g2_delta": np.zeros((1, G2W), np.uint64), "g2_xi": np.zeros((n, G2W), np.uint64), "gt_alpha_beta": np.zeros((1, FQ12), np.uint64)}
    c = Crs(n=n, l=l, m=m)
    for k, v in bufs.items(): setattr(c, k, ptr(v))
    return c, bufs


def dense(polys, n):
    return ints_to_arr([c for p in polys for c in (p + [0] * (n - len(p)))], 4)
