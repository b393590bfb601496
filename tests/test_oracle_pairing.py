"""Oracle pairing checks: the reference asserts only properties for pairings
(pairing.rs:107-213: bilinearity for weil and tate, e(2P+2P,Q)=e(2P,Q)^2) —
no numeric GT value exists in the reference, so GT bits are pinned transitively
by the tower/group KATs + these properties.  CPU only."""
import ctypes
import numpy as np
from zkt_testlib import *
from test_oracle_kats import g1_gen, g2_gen, g1_mul, g2_mul, g1_add, O


def pair(which, p, q, threads=8):
    n = p.shape[0]
    o = np.zeros((n, FQ12), dtype=np.uint64)
    idx = ctypes.c_size_t(0)
    rc = O.zkto_pairing_batch(which, ptr(p), ptr(q), ptr(o), n, threads, ctypes.byref(idx))
    return rc, o, idx.value


def gt_mul(a, b):
    o = np.zeros((1, FQ12), dtype=np.uint64)
    assert O.zkto_fq12_op(2, ptr(a), ptr(b), ptr(o), 1) == 0
    return o


def test_pairing_params():
    lb = np.zeros(300, dtype=np.uint32); fe = np.zeros(200, dtype=np.uint32)
    n1, n2 = ctypes.c_size_t(), ctypes.c_size_t()
    u32p = ctypes.POINTER(ctypes.c_uint32)
    assert O.zkto_pairing_params(lb.ctypes.data_as(u32p), ctypes.byref(n1), fe.ctypes.data_as(u32p), ctypes.byref(n2)) == 0
    bits = [int(b) for b in lb[:n1.value]]
    ref_bits = [int(c) for c in bin(R - 1)[3:]]            # pairing.rs:58-73: MSB first, leading 1 dropped
    assert bits == ref_bits and len(bits) == 254 and sum(bits) == 132
    exp = sum(int(v) << (32 * i) for i, v in enumerate(fe[:n2.value]))
    assert exp == (Q**12 - 1) // R and exp.bit_length() == 4314 and bin(exp).count("1") == 2124


def test_tate_and_weil_bilinearity_generators():   # pairing.rs:107-123, 136-151
    g1, g2 = g1_gen(), g2_gen()
    p10 = g1_mul(g1, 10); p11 = g1_add(g1, p10)
    ps = np.concatenate([g1, p10, p11]); qs = np.concatenate([g2, g2, g2])
    for which in (3, 2):
        rc, o, _ = pair(which, ps, qs)
        assert rc == 0
        assert (gt_mul(o[0:1], o[1:2]) == o[2:3]).all()
        assert not (o[0] == o[1]).all()


def test_tate_square_identity_random():            # pairing.rs:153-171,193-196 with seeded points
    rng = SplitMix64(2)
    p = g1_mul(g1_gen(), rng.below(R)); q = g2_mul(g2_gen(), rng.below(R))
    p2 = g1_add(p, p); p4 = g1_add(p2, p2)
    rc, o, _ = pair(3, np.concatenate([p2, p4]), np.concatenate([q, q]))
    assert rc == 0 and (gt_mul(o[0:1], o[0:1]) == o[1:2]).all()


def test_tate_generators_matches_survey_model_value():
    """value of tate(G1::g(), G2::g()) from an independent python model of
    pairing.rs:20-100 (model-derived, not a reference run) — cross-check, in to_strs order."""
    exp = [
        0x1392591849b6e2e1ad7e15c9f7ad7d006f3114a309a55da98ce84536baac51ddad746ebd6e3e7a42621a26c48473ee0a,
        0x1379a297d6f4ba1fdf0787f66df5f6cef4e2665fc2018f467a6bff3448329496d15f3df15fa17b171e069e57fd685d85,
        0x0711c138d3b30afe3bb848bd27460c35477621a85202ca265da9294ee5c32ce97b3f3f8db8ebfa7ef9c7c6532aeef9f6,
        0x11bc2a922801b36bf4672048c30f568f7e1dcf1659d4cbe90bdac5359a133603f80060baf2ab17599f03819f2976eb48,
        0x173f5527dca3373e41606580c134cdb6394fa852b9f2d2f2e49860426d9f8f3883219ccaa52474e312bb3137933ccfc8,
        0x07c318c63329f56711057e83c80afb106434cf15876eef5ffe0aeb3658e32ff809b0f23cbe122e81c71a486d96487f48,
        0x0d09739da55f5614f7c4d4797d1e03b3eb1535eec27289eeb7ce1b184790569aa3008b4fa90fde3f20a833077566e6b7,
        0x04afea0dfb2d1a8c6bd0e026229a00f6e528c6571b396520e0f2afacf65a897114bbc47a3a70a1dde7eff25f03e1b7c4,
        0x11c47f674c1e8b3ed34c10a737d3a1c8546020228974c42498f70485e31807f51db8b2603b7daa3ee51cec5b324a5a47,
        0x02b0ab8352b118d9abb1e8a4db2cfb4acc4dc84d8251959d32968f5c8d51d482913e84a3d56800407eb5c5fedaa4a751,
        0x10f46281da7f33fcd0328586a73d0843e708abb33400bae5b1f0df0ed639533f3f95835519b16ab62dc2da82b536a870,
        0x11cc3b83f86bbeca92000eb7896abd6070099f7aa9ab795ceebfdb579b02a4bfa51d8230627f6503e8f5600f770e3b41,
    ]
    rc, o, _ = pair(3, g1_gen(), g2_gen(), threads=1)
    assert rc == 0 and list(fq12_from_arr(o)[0]) == exp


def test_pairing_with_infinity_is_error():         # rational_function.rs:36,59 panics
    rc, _, idx = pair(3, np.concatenate([g1_gen(), g1_arr([None])]), np.concatenate([g2_gen(), g2_gen()]), threads=1)
    assert rc == ZKT_ERR_INFINITY and idx == 1
    rc, _, idx = pair(3, g1_gen(), g2_arr([None]), threads=1)
    assert rc == ZKT_ERR_INFINITY and idx == 0
