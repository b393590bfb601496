"""The exact math headers the HIP kernels are built from (fp.h, tower.h, curve.h,
pairing.h), compiled for the host, checked against the oracle.  CPU only.
This validates formulas and constants; the GPU parity tests validate the device build."""
import ctypes, os, subprocess
import numpy as np
import pytest
from zkt_testlib import *
from test_oracle_kats import g1_gen, g2_gen, g1_mul, g2_mul, g1_add, g2_add, sp_arr, O
from test_oracle_pairing import pair

_u32p = ctypes.POINTER(ctypes.c_uint32)


def p32(a):
    return a.ctypes.data_as(_u32p) if a is not None else None


# two host builds of the same headers: the default Fq2 product (four product scans) and -DZKT_FQ2_KARATSUBA (three), which the
# pairing objects of the product library are compiled with (zk-toolkit_amd/Makefile)
@pytest.fixture(scope="module", params=["plain", "kara"])
def H(request):
    if os.environ.get("ZKT_HOSTCHECK_SO"):          # e.g. a -fsanitize=undefined build of csrc/hostcheck.cpp
        if request.param != "plain":
            pytest.skip("one externally supplied build")
        return ctypes.CDLL(os.environ["ZKT_HOSTCHECK_SO"])
    kara = request.param == "kara"
    so = os.path.join(ROOT, "zk-toolkit_amd", "libzkt_hostcheck_kara.so" if kara else "libzkt_hostcheck.so")
    src = os.path.join(ROOT, "zk-toolkit_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src) if f.endswith(".h") or f == "hostcheck.cpp")      # what the host build includes (Makefile: $(HDRS))
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.check_call(["hipcc", "-x", "hip", "--cuda-host-only", "-O2", "-std=c++17", "-fPIC", "-shared"] + (["-DZKT_FQ2_KARATSUBA"] if kara else []) +
                              ["-o", so, os.path.join(src, "hostcheck.cpp")], timeout=600)
    return ctypes.CDLL(so)


def test_fq2_products_agree_on_lazy_representatives(H):
    """fp2_mul_kara (three product scans, fp.h) against the four-scan product and against python, operands at canonical + {0,1,2}p."""
    rng = SplitMix64(77)
    vals = [0, 1, Q - 1, Q - 2, (Q + 1) // 2, 2**380, 2**381 - 1 - (2**381 - 1) // Q * Q] + [rng.below(Q) for _ in range(40)]
    for t in range(400):
        a = (vals[rng.below(len(vals))], vals[rng.below(len(vals))]); b = (vals[rng.below(len(vals))], vals[rng.below(len(vals))])
        lifts = t % 256 if t < 256 else rng.below(256)
        got = np.zeros((1, 12), dtype=np.uint64)
        enc = lambda z: ints_to_arr([z[1], z[0]], 6).reshape(1, 12)          # ABI order {u1, u0}
        assert H.zkt_hostcheck_fq2_mul_lifted(p32(enc(a)), p32(enc(b)), lifts, p32(got)) == 0, (t, lifts)
        c1, c0 = arr_to_ints(got.reshape(2, 6))
        assert (c0, c1) == ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q), (t, lifts)


@pytest.mark.parametrize("field,mod,w,pre", [(0, Q, 6, "fq"), (1, R, 4, "fr"), (2, SECP_P, 4, "sp"), (3, SECP_N, 4, "sn")])
def test_field_ops(H, field, mod, w, pre):
    rng = SplitMix64(31 + field)
    xs = [rng.below(mod) for _ in range(64)] + [0, 1, mod - 1, mod - 1, 2]
    ys = [rng.below(mod) for _ in range(64)] + [0, mod - 1, mod - 1, 1, (mod + 1) // 2]
    a, b = ints_to_arr(xs, w), ints_to_arr(ys, w)
    for op, f in ((0, lambda x, y: (x + y) % mod), (1, lambda x, y: (x - y) % mod), (2, lambda x, y: x * y % mod),
                  (3, lambda x, y: x * x % mod), (4, lambda x, y: -x % mod)):
        o = np.zeros_like(a)
        assert H.zkt_hostcheck_fp(field, op, p32(a), p32(b), p32(o), len(xs)) == 0
        assert arr_to_ints(o) == [f(x, y) for x, y in zip(xs, ys)], op
    nz = ints_to_arr([x for x in xs if x] + [2**k for k in (1, 31, 32, 33, 64, 200)] + [mod - 2**k for k in (0, 1, 32, 100)], w)
    o = np.zeros_like(a)                               # a3: cube, pow with one exponent per element (fp_pow, the kernels' routine)
    assert H.zkt_hostcheck_fp(field, 7, p32(a), None, p32(o), len(xs)) == 0
    assert arr_to_ints(o) == [pow(x, 3, mod) for x in xs]
    es = [rng.below(mod) for _ in range(len(xs) - 3)] + [0, 1, mod - 1]
    assert H.zkt_hostcheck_fp(field, 8, p32(a), p32(ints_to_arr(es, w)), p32(o), len(xs)) == 0
    assert arr_to_ints(o) == [pow(x, e, mod) for x, e in zip(xs, es)]
    for op in (5, 6, 9, 10):                           # fp_inv (word-step binary GCD), x^(p-2) through fp_pow, the classic bit-step Euclid, the word-step GCD on canonical words
        o = np.zeros_like(nz)
        assert H.zkt_hostcheck_fp(field, op, p32(nz), None, p32(o), len(nz)) == 0
        assert arr_to_ints(o) == [pow(x, -1, mod) for x in arr_to_ints(nz)], op
    # the word-step GCD on inputs that stress its approximations: tiny values, values around 2^31 / 2^62 / word boundaries, p - small, (p +- 1) / 2,
    # all-ones patterns and 3000 random residues (a wrong factor matrix would leave b != 1 and take the fallback: results are compared with python's pow)
    edge = [1, 2, 3, mod - 1, mod - 2, (mod - 1) // 2, (mod + 1) // 2, 2**31 - 1, 2**31, 2**31 + 1, 2**62 - 1, 2**62, 2**63, 2**64 - 1, 2**64, 2**64 + 1]
    edge += [2**k for k in range(1, mod.bit_length() - 1, 7)] + [mod - 2**k for k in range(1, mod.bit_length() - 2, 11)] + [(1 << k) - 1 for k in range(2, mod.bit_length() - 1, 13)]
    edge += [rng.below(mod) for _ in range(3000)] + [rng.below(2**64) + 1 for _ in range(200)] + [mod - 1 - rng.below(2**64) for _ in range(200)]
    e = ints_to_arr([x % mod for x in edge if x % mod], w); o = np.zeros_like(e)
    assert H.zkt_hostcheck_fp(field, 10, p32(e), None, p32(o), len(e)) == 0
    assert arr_to_ints(o) == [pow(x, -1, mod) for x in arr_to_ints(e)]
    H.zkt_hostcheck_bgcd_fallbacks.restype = ctypes.c_ulong
    assert H.zkt_hostcheck_bgcd_fallbacks() == 0, "the word-step GCD fell back to the classic loop"


def _fq_program_model(seed, steps, regs):
    st = seed
    r = list(regs)
    for _ in range(steps):
        st = (st * 6364136223846793005 + 1442695040888963407) & (2**64 - 1)
        op, d, a, b = (st >> 33) % 16, (st >> 40) & 3, (st >> 42) & 3, (st >> 44) & 3
        if op in (0, 1): r[d] = (r[a] + r[b]) % Q
        elif op in (2, 3): r[d] = (r[a] - r[b]) % Q
        elif op == 4: r[d] = -r[a] % Q
        elif op == 5: r[d] = 2 * r[a] % Q
        elif op == 6: r[d] = r[a] * r[b] % Q
        elif op == 7: r[d] = r[a] * r[a] % Q
        elif op == 8: r[d] = r[a]
        elif op == 10: r[d] = (r[a] - r[b] - 2 * r[(b + 1) & 3]) % Q
        elif op == 11: r[d] = (r[a] * r[b] - r[(a + 1) & 3] * r[(b + 2) & 3]) % Q
        elif op == 12: r[d] = (r[a] * r[b] + r[(a + 1) & 3] * r[(b + 2) & 3]) % Q
        elif op == 13: r[d] = (r[a] + r[b] + r[(b + 1) & 3]) % Q
        elif op == 14: r[d] = (r[a] + r[b] - r[(b + 1) & 3]) % Q
        elif op == 15: r[d] = (r[a] - r[b] - r[(b + 1) & 3]) % Q
        else: r[d] = 1 if r[a] == r[b] else (r[a] + 1) % Q
    return r


@pytest.mark.parametrize("seed", range(12))
def test_fq_lazy_limb_programs(H, seed):
    """Fq is held in 14 lazily-reduced 28-bit limbs (< 4p): long random op chains, non-canonical representatives and
    boundary operands must keep the limb/size invariants and give the residues plain mod-p arithmetic gives."""
    rng = SplitMix64(900 + seed)
    edge = [0, 1, Q - 1, Q - 2, 2, (Q + 1) // 2, (1 << 380), (1 << 364) - 1, Q - (1 << 364), (1 << 28) - 1, Q >> 1]
    regs = [edge[(seed + i) % len(edge)] if (seed + i) % 3 == 0 else rng.below(Q) for i in range(4)]
    if seed == 1: regs = [0, 0, Q - 1, 1]
    if seed == 2: regs = [Q - 1, Q - 1, Q - 1, Q - 1]
    a = ints_to_arr(regs, 6); o = np.zeros_like(a)
    H.zkt_hostcheck_fq_program.argtypes = [ctypes.c_uint64, ctypes.c_int, _u32p, _u32p]
    steps = 4000
    assert H.zkt_hostcheck_fq_program(seed * 7919 + 1, steps, p32(a), p32(o)) == 0        # no invariant violation
    assert arr_to_ints(o) == _fq_program_model(seed * 7919 + 1, steps, regs)


def test_fq_lazy_reduce_quotient_estimate():
    """fp_lazy_reduce's q = floor(top * QEST_M / 2^32) from the top limb alone: never above floor(v/p) and at most 3
    below it for every v < 12p (so a reduced value is < 4p).  Checked at the extremes of every reachable top limb."""
    top_p = Q >> 364
    M = (1 << 32) // (top_p + 1)
    for top in range(0, (20 * Q >> 364) + 1):                 # fp_sub2 reduces values up to a + 16p < 20p
        q = (top * M) >> 32
        # un-normalised input: the limbs below the top may each hold up to 6*2^28 (fp_sub2), i.e. spill < 7 units into the top
        lo, hi = top << 364, min(((top + 7) << 364) - 1, 20 * Q - 1)
        assert q * Q <= lo                                 # never overshoots: v - q*p >= 0
        assert hi - q * Q < 4 * Q                          # result < 4p
    # add: operands < 4p -> v < 8p; sub: a + 8p - b < 12p, limbs stay below 2^31 (fp.h bounds)
    assert (20 * Q >> 364) * M < 2**63


@pytest.mark.parametrize("field,mod", [(2, SECP_P), (3, SECP_N)])
def test_field_ops_256bit_moduli(H, field, mod):
    rng = SplitMix64(41 + field)
    xs = [rng.below(mod) for _ in range(64)] + [mod - 1, mod - 1, 1]
    ys = [rng.below(mod) for _ in range(64)] + [mod - 1, 1, mod - 1]
    a, b = ints_to_arr(xs, 4), ints_to_arr(ys, 4)
    for op, f in ((0, lambda x, y: (x + y) % mod), (1, lambda x, y: (x - y) % mod), (2, lambda x, y: x * y % mod)):
        o = np.zeros_like(a)
        assert H.zkt_hostcheck_fp(field, op, p32(a), p32(b), p32(o), len(xs)) == 0
        assert arr_to_ints(o) == [f(x, y) for x, y in zip(xs, ys)], op
    for op in (5, 6):
        o = np.zeros_like(a)
        assert H.zkt_hostcheck_fp(field, op, p32(a), None, p32(o), len(xs)) == 0
        assert arr_to_ints(o) == [pow(x, -1, mod) for x in xs], op


def _rand_tower(rng, w):
    return ints_to_arr([rng.below(Q) for _ in range(w // 6)], 6).reshape(1, w)


@pytest.mark.parametrize("deg,w,fn,ops", [(2, 12, "zkto_fq2_op", (0, 1, 2, 3, 4, 5, 6)), (6, 36, "zkto_fq6_op", (0, 1, 2, 3, 4, 5)),
                                          (12, 72, "zkto_fq12_op", (0, 1, 2, 3, 4))])
def test_tower_vs_oracle(H, deg, w, fn, ops):
    rng = SplitMix64(50 + deg)
    for _ in range(3):
        a, b = _rand_tower(rng, w), _rand_tower(rng, w)
        for op in ops:
            want = np.zeros((1, w), dtype=np.uint64); got = np.zeros((1, w), dtype=np.uint64)
            assert getattr(O, fn)(op, ptr(a), ptr(b), ptr(want), 1) == 0
            assert H.zkt_hostcheck_tower(deg, op, p32(a), p32(b), p32(got)) == 0
            assert (want == got).all(), (deg, op)


def test_fq12_sqr_frobenius_conj(H):
    rng = SplitMix64(60)
    a = _rand_tower(rng, 72)
    want = np.zeros((1, 72), dtype=np.uint64); got = np.zeros((1, 72), dtype=np.uint64)
    assert O.zkto_fq12_op(2, ptr(a), ptr(a), ptr(want), 1) == 0
    assert H.zkt_hostcheck_tower(12, 6, p32(a), None, p32(got)) == 0 and (want == got).all()
    u32 = lambda v: np.array([(v >> (32 * i)) & 0xFFFFFFFF for i in range((v.bit_length() + 31) // 32)], dtype=np.uint32)
    for op, e in ((7, Q), (8, Q * Q), (9, Q**6)):
        ev = u32(e)
        assert O.zkto_fq12_pow(ptr(a), p32(ev), len(ev), ptr(want)) == 0
        assert H.zkt_hostcheck_tower(12, op, p32(a), None, p32(got)) == 0
        assert (want == got).all(), op


def test_groups_vs_oracle(H):
    rng = SplitMix64(70)
    g1, g2 = g1_gen(), g2_gen()
    secp = np.zeros((1, 9), dtype=np.uint64); O.zkto_secp_generator(ptr(secp))
    for grp, g, W, omul, oadd in ((0, g1, G1W, O.zkto_g1_mul_batch, O.zkto_g1_add_batch),
                                  (1, g2, G2W, O.zkto_g2_mul_batch, O.zkto_g2_add_batch),
                                  (2, secp, 9, O.zkto_secp_mul_batch, O.zkto_secp_add_batch)):
        order = SECP_N if grp == 2 else R
        def mul(p, k):
            s = ints_to_arr([k], 4); o = np.zeros((1, W), dtype=np.uint64)
            assert omul(ptr(p), ptr(s), 4, ptr(o), 1, 1) == 0; return o
        ks = [rng.below(order), 1, 2, 0, order, order - 1, order + 5, (1 << 256) - 1]
        for k in ks:
            s = ints_to_arr([k], 4); got = np.zeros((1, W), dtype=np.uint64)
            assert H.zkt_hostcheck_group(grp, 2, p32(g), p32(s), 8, p32(got)) == 0
            assert (got == mul(g, k)).all(), (grp, k)
        p, q = mul(g, rng.below(order)), mul(g, rng.below(order))
        inf = np.zeros((1, W), dtype=np.uint64); inf[0, W - 1] = 1
        negp = mul(p, order - 1)
        for a, b in ((p, q), (p, p), (p, negp), (p, inf), (inf, q), (inf, inf)):
            want = np.zeros((1, W), dtype=np.uint64)
            assert oadd(ptr(a), ptr(b), ptr(want), 1) == 0
            for op in (0, 1):
                got = np.zeros((1, W), dtype=np.uint64)
                assert H.zkt_hostcheck_group(grp, op, p32(a), p32(b), 0, p32(got)) == 0
                assert (got == want).all(), (grp, op)


def test_tate_vs_oracle(H):
    rng = SplitMix64(80)
    ps = np.concatenate([g1_gen(), g1_mul(g1_gen(), rng.below(R))])
    qs = np.concatenate([g2_gen(), g2_mul(g2_gen(), rng.below(R))])
    rc, want, _ = pair(3, ps, qs)
    assert rc == 0
    for i in range(2):
        got = np.zeros((1, 72), dtype=np.uint64)
        assert H.zkt_hostcheck_tate(p32(ps[i:i + 1]), p32(qs[i:i + 1]), p32(got)) == 0
        assert (got[0] == want[i]).all()
    assert H.zkt_hostcheck_tate(p32(g1_arr([None])), p32(qs[0:1]), p32(np.zeros((1, 72), dtype=np.uint64))) == 2


def test_pairing_with_g1_argument_outside_the_subgroup(H):
    """weak #8 of the round-1 review: for P outside G1 the reference's chain may meet infinity and panic (rational_function.rs:36), or
    return a value that depends on P's order.  The engine detects r P != infinity at the end of its fast loop and re-runs such elements
    on the reference's own chain: same value, or the same panic, as the oracle (which restates the panics as errors)."""
    q = g2_mul(g2_gen(), 777)
    for label, pt in degenerate_g1_points():
        p = g1_arr([pt])
        rc, want, _ = pair(3, p, q, threads=1)
        got = np.zeros((1, 72), dtype=np.uint64)
        hrc = H.zkt_hostcheck_tate(p32(p), p32(q), p32(got))
        if rc != 0:
            assert hrc == 2, (label, "the oracle reports the reference's panic, the engine returned", hrc)
        else:
            assert hrc == 100 and (got[0] == want[0]).all(), label           # 100: produced by the exact path
        rc, want, _ = pair(0, p, q, threads=1)                               # raw Miller value, same rule
        hrc = H.zkt_hostcheck_miller_exact(0, p32(p), p32(q), p32(got))
        assert (hrc == 2) if rc != 0 else (hrc == 0 and (got[0] == want[0]).all()), label


def test_short_loop_guards_and_fallbacks(H):
    """The 127-step loop (pairing.h) runs only for P on E with r P = infinity and Q in G2; everything else falls back to the 255-step loop (any Q)
    or the reference's chain (P outside G1).  Guards against the python model / plain scalar multiplication, values against the oracle."""
    rng = SplitMix64(91)
    H.zkt_hostcheck_short_loop_guards.argtypes = [_u32p, _u32p]
    p_ok = g1_mul(g1_gen(), rng.below(R)); q_ok = g2_mul(g2_gen(), rng.below(R))
    assert H.zkt_hostcheck_short_loop_guards(p32(p_ok), p32(q_ok)) == 15
    for label, pt in degenerate_g1_points():                                  # on the curve, outside G1
        assert H.zkt_hostcheck_short_loop_guards(p32(g1_arr([pt])), p32(q_ok)) == 13, label
    px, py_ = g1_from_arr(p_ok)[0]
    p_off = g1_arr([(px, (py_ + 1) % Q)])
    assert H.zkt_hostcheck_short_loop_guards(p32(p_off), p32(q_ok)) & 1 == 0          # off the curve
    (x1, x0), (y1, y0) = g2_from_arr(q_ok)[0]
    cases = [(g2_arr([to_abi_g2(py_twist_point(rng))]), 11, 50) for _ in range(2)]          # on E', outside G2: the 255-step loop
    cases.append((g2_arr([((x1, x0), (y1, (y0 + 1) % Q))]), 3, 100))                          # off the twist: the reference's chain
    for q, guards, route in cases:
        assert H.zkt_hostcheck_short_loop_guards(p32(p_ok), p32(q)) == guards
        rc, want, _ = pair(3, p_ok, q, threads=1)
        got = np.zeros((1, 72), dtype=np.uint64)
        assert rc == 0 and H.zkt_hostcheck_tate(p32(p_ok), p32(q), p32(got)) == route and (got[0] == want[0]).all(), route
    rc, want, _ = pair(3, p_off, q_ok, threads=1)                             # P off its curve: the reference's chain, value or panic
    got = np.zeros((1, 72), dtype=np.uint64)
    hrc = H.zkt_hostcheck_tate(p32(p_off), p32(q_ok), p32(got))
    assert (hrc == 2) if rc != 0 else (hrc == 100 and (got[0] == want[0]).all())


def test_exact_miller_and_weil_vs_oracle(H):         # pairing.rs:54-55,75-84 raw values
    rng = SplitMix64(81)
    p = g1_mul(g1_gen(), rng.below(R)); q = g2_mul(g2_gen(), rng.below(R))
    for which in (0, 1, 2):
        rc, want, _ = pair(which, p, q, threads=1)
        assert rc == 0
        got = np.zeros((1, 72), dtype=np.uint64)
        assert H.zkt_hostcheck_miller_exact(which, p32(p), p32(q), p32(got)) == 0
        assert (got[0] == want[0]).all(), which


def test_cyclotomic_square_matches_generic_square(H):
    """Granger-Scott squaring == plain squaring on elements of the cyclotomic subgroup (a^((q^6-1)(q^2+1)))."""
    rng = SplitMix64(61)
    u32 = lambda v: np.array([(v >> (32 * i)) & 0xFFFFFFFF for i in range((v.bit_length() + 31) // 32)], dtype=np.uint32)
    e = u32((Q**6 - 1) * (Q**2 + 1))
    for _ in range(2):
        a = _rand_tower(rng, 72); c = np.zeros((1, 72), dtype=np.uint64)
        assert O.zkto_fq12_pow(ptr(a), p32(e), len(e), ptr(c)) == 0
        want = np.zeros((1, 72), dtype=np.uint64); got = np.zeros((1, 72), dtype=np.uint64)
        assert O.zkto_fq12_op(2, ptr(c), ptr(c), ptr(want), 1) == 0
        assert H.zkt_hostcheck_tower(12, 10, p32(c), None, p32(got)) == 0
        assert (want == got).all()


def test_ate_product_matches_model_and_guards(H):
    """The 63-step loop of the deciding entry points (pairing.h miller_ate_multi, its 13-product line multiplication, the line tables of shared
    G2 points, g1_in_subgroup): final_exponentiation of the product equals the python model's value bit for bit — with the chain in the lane and with
    tabulated lines — and arguments outside their groups are refused (they keep the older routes)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fast_model as fm
    H.zkt_hostcheck_ate_product.argtypes = [ctypes.c_int, ctypes.c_int, _u32p, _u32p, _u32p]
    rng = SplitMix64(93)
    a, b, c = (rng.below(R) for _ in range(3))
    g1s = [g1_mul(g1_gen(), a), g1_mul(g1_gen(), R - c), g1_gen()]
    g2s = [g2_mul(g2_gen(), b), g2_gen(), g2_mul(g2_gen(), (c - a * b) % R)]
    def model(idx):
        ps = [g1_from_arr(g1s[i])[0] for i in idx]
        qs = []
        for i in idx:
            (x1, x0), (y1, y0) = g2_from_arr(g2s[i])[0]; qs.append(((x0, x1), (y0, y1)))
        return fm.ate_product(ps, qs)
    one = np.zeros(72, dtype=np.uint64); one[66] = 1                          # canonical one: w0.v0.u0 (the last Fq of the {w1,w0} layout)
    for kv, kf, idx in ((1, 0, [0]), (0, 1, [0]), (2, 0, [0, 1]), (3, 0, [0, 1, 2]), (1, 2, [0, 1, 2])):
        got = np.zeros((1, 72), dtype=np.uint64)
        P = np.concatenate([g1s[i] for i in idx]); Qa = np.concatenate([g2s[i] for i in idx])
        assert H.zkt_hostcheck_ate_product(kv, kf, p32(P), p32(Qa), p32(got)) == 0, (kv, kf)
        want = fm.to_ref_order(model(idx))
        assert tuple(fq12_from_arr(got)[0]) == tuple(want), (kv, kf)
        assert (got[0] == one).all() == (len(idx) == 3)                        # the three-pair product is one by construction
    got = np.zeros((1, 72), dtype=np.uint64)
    q_ok = g2s[0]
    for label, pt in degenerate_g1_points():
        assert H.zkt_hostcheck_ate_product(1, 0, p32(g1_arr([pt])), p32(q_ok), p32(got)) == -1, label
    tw = g2_arr([to_abi_g2(py_twist_point(rng))])
    assert H.zkt_hostcheck_ate_product(1, 0, p32(g1s[0]), p32(tw), p32(got)) == -1                   # on E', outside G2: caught where the chain ends
    assert H.zkt_hostcheck_ate_product(0, 1, p32(g1s[0]), p32(tw), p32(got)) == -1                   # ... and by the table builder, agreeing with g2_in_subgroup (no bit 8)
    (x1, x0), (y1, y0) = g2_from_arr(q_ok)[0]
    assert H.zkt_hostcheck_ate_product(1, 0, p32(g1s[0]), p32(g2_arr([((x1, x0), (y1, (y0 + 1) % Q))])), p32(got)) == -1
