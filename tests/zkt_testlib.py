"""Shared helpers for the test-suite: ctypes handle on the CPU oracle (checker),
limb <-> int conversion in the include/zkt.h layouts, deterministic inputs.

TEST INFRASTRUCTURE.  The oracle is only ever used here as the checker."""
import ctypes, json, os, subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
SECP_P = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F
SECP_N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141

FQ, FR, G1W, G2W, FQ2, FQ6, FQ12 = 6, 4, 13, 25, 12, 36, 72
ZKT_OK, ZKT_ERR_INV_ZERO, ZKT_ERR_INFINITY, ZKT_ERR_SHAPE, ZKT_ERR_DEVICE = 0, 1, 2, 3, 4

_u64p = ctypes.POINTER(ctypes.c_uint64)


def ptr(a):
    return a.ctypes.data_as(_u64p) if a is not None else None


def kats():
    with open(os.path.join(GOLDEN, "ref_kats.json")) as f:
        return json.load(f)


# ---------------------------------------------------------------- oracle ----
_oracle = None


def oracle():
    """Build (if needed) and load oracle/libzkt_oracle.so."""
    global _oracle
    if _oracle is None:
        so = os.path.join(ROOT, "oracle", "libzkt_oracle.so")
        srcs = [os.path.join(ROOT, "oracle", f) for f in ("zkt_oracle.cpp", "zkt_oracle_capi.cpp", "zkt_oracle_protocols.cpp", "zkt_oracle.hpp")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        _oracle = ctypes.CDLL(so)
    return _oracle


# ------------------------------------------------------------ conversions ----
def int_to_limbs(x, n):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def limbs_to_int(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


def ints_to_arr(xs, n):
    """list of ints -> (len, n) u64 array."""
    a = np.zeros((len(xs), n), dtype=np.uint64)
    for i, x in enumerate(xs):
        a[i] = int_to_limbs(x, n)
    return a


def arr_to_ints(a):
    a = np.asarray(a)
    return [limbs_to_int(row) for row in a.reshape(-1, a.shape[-1])]


def g1_arr(points):
    """[(x,y) | None] -> (n,13) u64 in zkt_g1_affine layout."""
    a = np.zeros((len(points), G1W), dtype=np.uint64)
    for i, p in enumerate(points):
        if p is None:
            a[i, 12] = 1
        else:
            a[i, :6] = int_to_limbs(p[0], 6)
            a[i, 6:12] = int_to_limbs(p[1], 6)
    return a


def g1_from_arr(a):
    out = []
    for row in np.asarray(a).reshape(-1, G1W):
        out.append(None if int(row[12]) & 0xFFFFFFFF else (limbs_to_int(row[:6]), limbs_to_int(row[6:12])))
    return out


def g2_arr(points):
    """[((x1,x0),(y1,y0)) | None] -> (n,25) u64 in zkt_g2_affine layout ({u1,u0} order)."""
    a = np.zeros((len(points), G2W), dtype=np.uint64)
    for i, p in enumerate(points):
        if p is None:
            a[i, 24] = 1
        else:
            (x1, x0), (y1, y0) = p
            a[i, 0:6] = int_to_limbs(x1, 6); a[i, 6:12] = int_to_limbs(x0, 6)
            a[i, 12:18] = int_to_limbs(y1, 6); a[i, 18:24] = int_to_limbs(y0, 6)
    return a


def g2_from_arr(a):
    out = []
    for row in np.asarray(a).reshape(-1, G2W):
        if int(row[24]) & 0xFFFFFFFF:
            out.append(None)
        else:
            out.append(((limbs_to_int(row[0:6]), limbs_to_int(row[6:12])), (limbs_to_int(row[12:18]), limbs_to_int(row[18:24]))))
    return out


def fq12_from_arr(a):
    """(…,72) -> list of 12-tuples in the reference's to_strs order (fq12.rs:179-195)."""
    return [tuple(limbs_to_int(row[6 * j:6 * j + 6]) for j in range(12)) for row in np.asarray(a).reshape(-1, FQ12)]


# ------------------------------------------------------------------ RNG ----
class SplitMix64:
    """Deterministic input generator."""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def below(self, m):
        """prime_field.rs:73-85: ceil(bits/8) random big-endian bytes reduced mod order."""
        nbytes = (m.bit_length() + 7) // 8
        v = 0
        for _ in range((nbytes + 7) // 8):
            v = (v << 64) | self.next()
        return (v & ((1 << (8 * nbytes)) - 1)) % m


def rand_u64_array(seed, shape):
    """Bulk deterministic u64s (numpy PCG64 keyed by seed) for large inputs."""
    return np.random.Generator(np.random.PCG64(seed)).integers(0, 2**64, size=shape, dtype=np.uint64)


def rand_scalars(seed, n, modulus=R):
    """n uniform scalars < modulus as (n,4) u64 (rejection-free: 320-bit draw mod m)."""
    raw = rand_u64_array(seed, (n, 5))
    out = np.zeros((n, 4), dtype=np.uint64)
    for i in range(n):
        out[i] = int_to_limbs(limbs_to_int(raw[i]) % modulus, 4)
    return out


# ------------------------------------------------ tiny python group model ----
def py_g1_add(p, q):
    """Affine add on y^2=x^3+4 over python ints (independent of the C++ oracle)."""
    if p is None: return q
    if q is None: return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2 and (y1 + y2) % Q == 0: return None
    if p == q:
        m = 3 * x1 * x1 * pow(2 * y1, -1, Q) % Q
    else:
        m = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (m * m - x1 - x2) % Q
    return (x3, (m * (x1 - x3) - y1) % Q)


def py_g1_mul(p, k):
    r = None
    while k:
        if k & 1: r = py_g1_add(r, p)
        p = py_g1_add(p, p); k >>= 1
    return r


def py_secp_add(p, q):
    """Affine add on secp256k1 (y^2 = x^3 + 7) over python ints."""
    if p is None: return q
    if q is None: return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2 and (y1 + y2) % SECP_P == 0: return None
    m = (3 * x1 * x1 * pow(2 * y1, -1, SECP_P) if p == q else (y2 - y1) * pow(x2 - x1, -1, SECP_P)) % SECP_P
    x3 = (m * m - x1 - x2) % SECP_P
    return (x3, (m * (x1 - x3) - y1) % SECP_P)


def py_secp_mul(p, k):
    r = None
    while k:
        if k & 1: r = py_secp_add(r, p)
        p = py_secp_add(p, p); k >>= 1
    return r


SECP_GEN = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798, 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)


def secp_arr(points):
    a = np.zeros((len(points), 9), dtype=np.uint64)
    for i, p in enumerate(points):
        if p is None: a[i, 8] = 1
        else: a[i, :4] = int_to_limbs(p[0], 4); a[i, 4:8] = int_to_limbs(p[1], 4)
    return a


def _f2m(a, b): return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)
def _f2inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, Q)
    return (a[0] * d % Q, -a[1] * d % Q)


def py_g2_add(p, q):
    """Affine add on the twist y^2 = x^3 + 4(1+u) over Fq2 = Fq[u]/(u^2+1), python ints; points are ((x0,x1),(y0,y1)) in c0/c1 order."""
    if p is None: return q
    if q is None: return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2 and ((y1[0] + y2[0]) % Q, (y1[1] + y2[1]) % Q) == (0, 0): return None
    if p == q:
        xx = _f2m(x1, x1)
        m = _f2m((3 * xx[0] % Q, 3 * xx[1] % Q), _f2inv((2 * y1[0] % Q, 2 * y1[1] % Q)))
    else:
        m = _f2m(((y2[0] - y1[0]) % Q, (y2[1] - y1[1]) % Q), _f2inv(((x2[0] - x1[0]) % Q, (x2[1] - x1[1]) % Q)))
    mm = _f2m(m, m)
    x3 = ((mm[0] - x1[0] - x2[0]) % Q, (mm[1] - x1[1] - x2[1]) % Q)
    t = _f2m(m, ((x1[0] - x3[0]) % Q, (x1[1] - x3[1]) % Q))
    return (x3, ((t[0] - y1[0]) % Q, (t[1] - y1[1]) % Q))


def py_g2_mul(p, k):
    r = None
    while k:
        if k & 1: r = py_g2_add(r, p)
        p = py_g2_add(p, p); k >>= 1
    return r


G1_GEN = (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
          0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1)
G2_GEN = ((0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E,
           0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8),
          (0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE,
           0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801))


# ----------------------------------------- points outside the order-r subgroup ----
G1_COFACTOR = 0x396C8C005555E1568C00AAAB0000AAAB      # #E(Fq) = h * r, h = (z-1)^2 / 3


def py_g1_curve_point(seed):
    """a deterministic point of E(Fq): y^2 = x^3 + 4, generic order (both an r-component and a cofactor component)"""
    x = seed
    while True:
        rhs = (x * x * x + 4) % Q
        y = pow(rhs, (Q + 1) // 4, Q)                 # q = 3 mod 4
        if y * y % Q == rhs:
            return (x, y)
        x += 1


def degenerate_g1_points():
    """[(label, point)]: inputs on which the reference's Miller loop either panics or runs outside its intended domain"""
    gen = py_g1_curve_point(5)
    cof = py_g1_mul(gen, R)                           # r-component removed: order divides h
    assert cof is not None and py_g1_mul(cof, G1_COFACTOR) is None
    small = py_g1_mul(cof, G1_COFACTOR // (11 * 11))  # order divides 121
    assert small is not None and py_g1_mul(small, 121) is None
    mixed = py_g1_add(py_g1_mul(G1_GEN, 12345), cof)  # r-component and cofactor component
    return [("generic curve point", gen), ("cofactor subgroup, large order", cof), ("order | 121", small), ("order 3: (0, 2)", (0, 2)),
            ("G1 point + cofactor point", mixed)]


# ---- points of the twist E'(Fq2): y^2 = x^3 + 4(1+u), outside G2 -----------------------------------------------------------
def py_f2_sqrt(a):
    """square root in Fq2 = Fq[u]/(u^2+1) of a = (c0, c1), q = 3 mod 4; None if a is not a square"""
    a0, a1 = a
    if a1 == 0:
        s = pow(a0, (Q + 1) // 4, Q)
        if s * s % Q == a0: return (s, 0)
        s = pow(-a0 % Q, (Q + 1) // 4, Q)
        return (0, s) if s * s % Q == -a0 % Q else None
    n = (a0 * a0 + a1 * a1) % Q
    s = pow(n, (Q + 1) // 4, Q)
    if s * s % Q != n: return None
    for sg in (s, -s % Q):
        h = (a0 + sg) * pow(2, -1, Q) % Q
        x0 = pow(h, (Q + 1) // 4, Q)
        if x0 and x0 * x0 % Q == h:
            return (x0, a1 * pow(2 * x0, -1, Q) % Q)
    return None


def py_twist_point(rng):
    """a point ((x0,x1),(y0,y1)) (c0/c1 order) of E'(Fq2), generic order: almost surely outside G2 (the cofactor has 508 bits)"""
    f2m = lambda a, b: ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)
    while True:
        x = (rng.below(Q), rng.below(Q))
        x3 = f2m(f2m(x, x), x)
        y = py_f2_sqrt(((x3[0] + 4) % Q, (x3[1] + 4) % Q))
        if y is not None: return (x, y)


def to_abi_g2(pt):
    """((x0,x1),(y0,y1)) -> ((x1,x0),(y1,y0)), the order g2_arr takes"""
    (x0, x1), (y0, y1) = pt
    return ((x1, x0), (y1, y0))


BLS_X = -0xD201000000010000
G2_COFACTOR = (BLS_X**8 - 4 * BLS_X**7 + 5 * BLS_X**6 - 4 * BLS_X**4 + 6 * BLS_X**3 - 4 * BLS_X**2 - 4 * BLS_X + 13) // 9      # #E'(Fq2) = h2 * r
