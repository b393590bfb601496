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
    # gates as Gate::build emits them (gate.rs:206-235): t1 = x*x, t2 = x*t1, t3 = (x+5)*1, t4 = (t2+t3)*1, out = t4*1
    A = [[0, 1, 0, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0, 0], [5, 1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 1, 1, 0], [0, 0, 0, 0, 0, 0, 1]]
    B = [[0, 1, 0, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0]]
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


def bits_circuit(n, seed=7):
    """A witness of 0/1 values (what a real circuit's witness is full of): n-1 booleanity constraints b_j * b_j = b_j and one packing
    constraint (sum_j 2^j b_j) * 1 = out; wires [one, out | b_0 .. b_{n-2}], l = 1.  (A w) and (B w) are 0/1 vectors: the MSM's skew path."""
    from zkt_testlib import SplitMix64
    rng = SplitMix64(seed)
    bs = [rng.below(2) for _ in range(n - 1)]
    out = sum(b << j for j, b in enumerate(bs)) % R
    wit = [1, out] + bs
    cols = n + 1
    A = [[0] * cols for _ in range(n)]; B = [[0] * cols for _ in range(n)]; C = [[0] * cols for _ in range(n)]
    for j in range(n - 1):
        A[j][2 + j] = 1; B[j][2 + j] = 1; C[j][2 + j] = 1
    for j in range(n - 1): A[n - 1][2 + j] = pow(2, j, R)
    B[n - 1][0] = 1; C[n - 1][1] = 1
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


# ---- evaluation-domain form of the same QAP: what zkt_groth16_setup_r1cs / prove_r1cs compute ----
def sparse_rows(M):
    """dense n x cols matrix -> CSR (rowptr u64[n+1], col u32[nnz], val (nnz,4) u64)"""
    rowptr, col, val = [0], [], []
    for row in M:
        for i, v in enumerate(row):
            if v % R: col.append(i); val.append(v % R)
        rowptr.append(len(col))
    return np.array(rowptr, np.uint64), np.array(col if col else [0], np.uint32), ints_to_arr(val if val else [0], 4)


def domain_model(Amat, Bmat, Cmat, witness, x):
    """Python-int model of the scalable prover's Fr stage on the reference's domain {1..n}:
       returns (a(x), b(x), h(x)*t(x)) computed WITHOUT any coefficient-form polynomial:
       a(x) = sum_j (A w)_j L_j(x);  h on E = {n+1..2n-1} through the arithmetic-progression shift (one convolution
       with 1/d), h(x) t(x) = t(x) * sum_s h(n+s) Lambda_s(x)."""
    n = len(Amat)
    fact = [1] * (2 * n + 1)
    for k in range(1, 2 * n + 1): fact[k] = fact[k - 1] * k % R
    inv = lambda v: pow(v, -1, R)
    tx = 1
    for j in range(1, n + 1): tx = tx * (x - j) % R
    tprime_inv = lambda j: (-1) ** (n - j) * inv(fact[j - 1] * fact[n - j]) % R
    L = [tx * inv((x - j) % R) * tprime_inv(j) % R for j in range(1, n + 1)]
    dot = lambda M: [sum(M[j][i] * witness[i] for i in range(len(witness))) % R for j in range(n)]
    Az, Bz, Cz = dot(Amat), dot(Bmat), dot(Cmat)
    ax = sum(v * l for v, l in zip(Az, L)) % R
    bx = sum(v * l for v, l in zip(Bz, L)) % R
    if n == 1: return ax, bx, 0
    def shift(vals):                    # values at 1..n -> values at n+1..2n-1
        f = [vals[j - 1] * tprime_inv(j) % R for j in range(1, n + 1)]
        out = []
        for s in range(1, n):
            S = sum(f[j - 1] * inv(n + s - j) for j in range(1, n + 1)) % R
            P = fact[n + s - 1] * inv(fact[s - 1]) % R
            out.append((S, P))
        return out
    sa, sb, sc = shift(Az), shift(Bz), shift(Cz)
    hE = [(P * Sa % R * Sb - Sc) % R for (Sa, P), (Sb, _), (Sc, _) in zip(sa, sb, sc)]     # (P Sa * P Sb - P Sc) / t(n+s), t(n+s) = P
    TE = 1
    for e in range(n + 1, 2 * n): TE = TE * (x - e) % R
    ht = 0
    for s in range(1, n):
        lam = TE * inv((x - (n + s)) % R) % R * ((-1) ** (n - 1 - s) * inv(fact[s - 1] * fact[n - 1 - s]) % R) % R
        ht = (ht + hE[s - 1] * lam) % R
    return ax, bx, ht * tx % R


class SparseRows(ctypes.Structure):
    """zkt_sparse_rows (include/zkt.h): one sparse row per constraint, as R1CS.constraints holds them (r1cs.rs, constraint.rs:5-9)"""
    _fields_ = [("rowptr", ctypes.POINTER(ctypes.c_uint64)), ("col", ctypes.POINTER(ctypes.c_uint32)), ("val", ctypes.POINTER(ctypes.c_uint64))]


def sparse_struct(rowptr, col, val):
    s = SparseRows(rowptr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), col.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ptr(val))
    s._keep = (rowptr, col, val)
    return s


def chain_circuit_sparse(n, seed=7):
    """chain_circuit(n) built directly in CSR form — usable at n = 2^20.  Returns (A, B, C) as
    (rowptr, col, val) triples, the witness as an (m+1, 4) u64 array, l = 1, m = n + 1."""
    from zkt_testlib import SplitMix64
    rng = SplitMix64(seed)
    cs = [rng.below(1 << 32) for _ in range(n)]
    ws = [rng.below(R)]
    for j in range(n): ws.append((ws[j] * ws[j] + cs[j]) % R)
    wit = [1, ws[n]] + ws[:n]
    one = np.array([1, 0, 0, 0], np.uint64)
    idx = np.arange(n, dtype=np.uint32) + 2                        # wire of w_j
    nxt = np.concatenate([idx[1:], np.array([1], np.uint32)])      # wire of w_{j+1} (the last one is `out`)
    rp1 = np.arange(n + 1, dtype=np.uint64)
    vA = np.tile(one, (n, 1))
    # C row j: (-c_j) * one + 1 * w_{j+1}; wire 0 first
    colC = np.empty(2 * n, np.uint32); colC[0::2] = 0; colC[1::2] = nxt
    valC = np.zeros((2 * n, 4), np.uint64)
    valC[0::2] = ints_to_arr([(-c) % R for c in cs], 4); valC[1::2] = one
    return ((rp1, idx.copy(), vA.copy()), (rp1.copy(), idx.copy(), vA.copy()), (np.arange(0, 2 * n + 1, 2, dtype=np.uint64), colC, valC)), ints_to_arr(wit, 4), 1, n + 1


# ---- Pinocchio (pinocchio/{crs,prover,verifier}.rs): ctypes mirrors of zkt_pinocchio_crs / zkt_pinocchio_proof -------------
_PIN_EK = [("vk_mid", G1W, "mid"), ("g1_wk_mid", G1W, "mid"), ("g2_wk_mid", G2W, "mid"), ("yk_mid", G1W, "mid"), ("alpha_vk_mid", G1W, "mid"),
           ("alpha_wk_mid", G1W, "mid"), ("alpha_yk_mid", G1W, "mid"), ("si", G2W, "deg"), ("beta_vwy_k_mid", G1W, "mid")]
_PIN_VK = [("one_g1", G1W, 1), ("one_g2", G2W, 1), ("alpha_v", G2W, 1), ("alpha_w", G1W, 1), ("alpha_y", G2W, 1), ("gamma", G2W, 1), ("beta_gamma", G2W, 1),
           ("t", G1W, 1), ("vk_io", G1W, "io"), ("wk_io", G2W, "io"), ("yk_io", G1W, "io"), ("alpha_v_t", G1W, 1), ("alpha_y_t", G1W, 1), ("beta_t", G1W, 1)]
_PIN_PROOF = [("v_mid_s", G1W), ("g1_w_mid_s", G1W), ("g2_w_mid_s", G2W), ("y_mid_s", G1W), ("h_s", G2W), ("alpha_v_mid_s", G1W), ("alpha_w_mid_s", G1W),
              ("alpha_y_mid_s", G1W), ("beta_vwy_mid_s", G1W)]


class PinCrs(ctypes.Structure):
    _fields_ = [(k, ctypes.c_size_t) for k in ("n", "n_io", "n_mid", "max_degree")] + [(k, ctypes.POINTER(ctypes.c_uint64)) for k, _, _ in _PIN_EK + _PIN_VK]


class PinProof(ctypes.Structure):
    _fields_ = [(k, ctypes.POINTER(ctypes.c_uint64)) for k, _ in _PIN_PROOF]


def alloc_pinocchio(n, n_io, n_mid, max_degree):
    cnt = {"mid": max(n_mid, 1), "io": max(n_io, 1), "deg": max_degree, 1: 1}
    bufs = {k: np.zeros((cnt[c], w), np.uint64) for k, w, c in _PIN_EK + _PIN_VK}
    crs = PinCrs(n=n, n_io=n_io, n_mid=n_mid, max_degree=max_degree)
    for k, v in bufs.items(): setattr(crs, k, ptr(v))
    return crs, bufs


def alloc_pinocchio_proof():
    bufs = {k: np.zeros((1, w), np.uint64) for k, w in _PIN_PROOF}
    pf = PinProof()
    for k, v in bufs.items(): setattr(pf, k, ptr(v))
    return pf, bufs


def pinocchio_instance(Amat, Bmat, Cmat, wit):
    """dense vi/wi/yi, quotient h and max_degree as Prover::new computes them (pinocchio/prover.rs:50-96)"""
    n = len(Amat)
    vi, wi, yi, h, t = qap_from_r1cs(Amat, Bmat, Cmat, wit)
    comb = lambda P: [sum(wit[i] * P[i][k] for i in range(len(wit))) % R for k in range(n)]
    p = poly_mul(comb(vi), comb(wi))
    for k, c in enumerate(comb(yi)): p[k] = (p[k] - c) % R
    deg = lambda q: max([k for k, c in enumerate(q) if c % R] + [0])
    max_degree = max([deg(q) for P in (vi, wi, yi) for q in P] + [deg(p), deg(t)]) + 1      # prover.rs:68-78
    return dense(vi, n), dense(wi, n), dense(yi, n), h, max_degree


def qap_util_pin_ek(): return list(_PIN_EK)
def qap_util_pin_vk(): return list(_PIN_VK)


def groth16_proof_scalars(mats, wires, l, trap, r, s):
    """The discrete logarithms of a Groth16 proof's (A, B, C) in python integers, O(n + nnz): with the trapdoor known,
         A = (alpha + a(x) + r delta) G1,   B = (beta + b(x) + s delta) G2,
         C = ((beta U_wit + alpha V_wit + W_wit + a(x) b(x) - c(x)) / delta + s A + r B - r s delta) G1      (prover.rs:96-147, crs.rs:65-121)
       where a(x) = sum_j (A w)_j L_j(x) over the Lagrange basis of {1..n} (qap.rs:33-97), U_wit = the same with the witness columns only, and
       h(x) t(x) = a(x) b(x) - c(x) for a satisfying witness (prover.rs:64-71).  One batch inversion for the n values x - j.
       mats: three CSR triples (rowptr, col, val[nnz,4] u64); wires: (m+1, 4) u64; trap = [alpha, beta, gamma, delta, x] as 1x4 u64 arrays.
       Independent of the HIP path AND of the oracle."""
    from zkt_testlib import R
    to_int = lambda a: [int.from_bytes(np.ascontiguousarray(row).tobytes(), "little") for row in np.asarray(a).reshape(-1, 4)]
    alpha, beta, gamma, delta, x = (to_int(t)[0] for t in trap)
    r, s = to_int(r)[0], to_int(s)[0]
    w = to_int(wires)
    n = len(mats[0][0]) - 1
    # L_j(x) = t(x) / ((x - j) t'(j)),  t'(j) = (-1)^(n-j) (j-1)! (n-j)!
    fact = [1] * (n + 1)
    for k in range(1, n + 1): fact[k] = fact[k - 1] * k % R
    d = [(x - j) % R for j in range(1, n + 1)]
    pre = [1] * (n + 1)
    for j in range(n): pre[j + 1] = pre[j] * d[j] % R
    tx = pre[n]
    assert tx != 0, "x lies in the domain"
    inv_run = pow(pre[n] * 1, -1, R)
    dinv = [0] * n
    for j in range(n - 1, -1, -1):
        dinv[j] = inv_run * pre[j] % R
        inv_run = inv_run * d[j] % R
    # 1 / ((j-1)! (n-j)!) by one more batch inversion
    den = [fact[j - 1] * fact[n - j] % R for j in range(1, n + 1)]
    pre2 = [1] * (n + 1)
    for j in range(n): pre2[j + 1] = pre2[j] * den[j] % R
    inv_run = pow(pre2[n], -1, R)
    Lx = [0] * n
    for j in range(n - 1, -1, -1):
        di = inv_run * pre2[j] % R
        inv_run = inv_run * den[j] % R
        v = tx * dinv[j] % R * di % R
        Lx[j] = v if (n - (j + 1)) % 2 == 0 else (R - v) % R
    def evals(M):
        rowptr, col, val = M
        rp = [int(v) for v in rowptr]; cl = [int(c) for c in col]; vl = to_int(val)
        full = wit = 0
        for j in range(n):
            fj = wj = 0
            for k in range(rp[j], rp[j + 1]):
                t = vl[k] * w[cl[k]]
                fj += t
                if cl[k] > l: wj += t
            full += fj % R * Lx[j]; wit += wj % R * Lx[j]
        return full % R, wit % R
    (a, U), (b, V), (c, W) = evals(mats[0]), evals(mats[1]), evals(mats[2])
    As = (alpha + a + r * delta) % R
    Bs = (beta + b + s * delta) % R
    dinv_ = pow(delta, -1, R)
    Cs = ((beta * U + alpha * V + W + a * b - c) * dinv_ + s * As + r * Bs - r * s * delta) % R
    return As, Bs, Cs
