#!/usr/bin/env python3
"""Transcribe the reference's own known-answer values into tests/golden/ref_kats.json.

The reference (exfinen/zk-toolkit, /root/reference) is Rust and cannot be built
in this image, so the CPU oracle is pinned by the numeric constants the
reference's inline #[test] modules assert.  This script pulls those *numbers*
(decimal / hex string literals) out of the given line ranges of the reference's
test modules and records them with file:line provenance.  Only data is
extracted — no source text is kept.  Run here (where /root/reference exists);
the JSON is committed and is what travels to the GPU box.
"""
import json, os, re, sys

REF = os.environ.get("ZKT_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_kats.json")

LIT = re.compile(r'b?"([0-9A-Fa-f]+)"')


def lits(rel, lo, hi):
    """All numeric string literals in lines [lo,hi] (1-based) -> [(line, text)]."""
    out = []
    with open(os.path.join(REF, rel)) as f:
        for i, line in enumerate(f, 1):
            if lo <= i <= hi:
                if line.lstrip().startswith("//"):
                    continue
                for m in LIT.finditer(line):
                    out.append((i, m.group(1)))
    return out


def ints(rel, lo, hi, base=10):
    return [int(t, base) for _, t in lits(rel, lo, hi)]


def u32_triples(rel, lo, hi):
    """InvTestCase { order: 97u32, n: 1u32, exp: 1u32 } rows."""
    pat = re.compile(r"order:\s*(\d+)u32,\s*n:\s*(\d+)u32,\s*exp:\s*(\d+)u32")
    out = []
    with open(os.path.join(REF, rel)) as f:
        for i, line in enumerate(f, 1):
            if lo <= i <= hi:
                m = pat.search(line)
                if m:
                    out.append([int(m.group(1)), int(m.group(2)), int(m.group(3))])
    return out


def main():
    k = {"_note": "numbers transcribed from the reference's #[test] modules; provenance = src file:line-range"}
    S = lambda v: [str(x) for x in v]  # big ints as decimal strings (JSON-safe)

    pfe = "src/building_block/field/prime_field_elem.rs"
    v = ints(pfe, 602, 612)
    k["fq_mul_large"] = {"src": pfe + ":601-617", "order": str(v[0]), "rhs": str(v[1]), "exp": str(v[2]), "lhs": "1234"}
    k["inv_small_primes"] = {"src": pfe + ":625-800", "cases": u32_triples(pfe, 625, 800)}
    v = lits(pfe, 811, 821)
    k["inv_secp256k1"] = {"src": pfe + ":811-821", "p_hex": v[0][1], "a": "1112121212121", "exp": str(int(v[1][1]))}
    k["pow_various"] = {"src": pfe + ":888-909", "order": "100000000",
                        "cases": [[2, 0, 1], [2, 1, 2], [2, 2, 4], [2, 3, 8], [3, 0, 1], [3, 1, 3], [3, 2, 9], [3, 3, 27], [17, 7, 10338673]]}

    # tower KATs; inputs are fq_test_helper.rs:9-34 (-3,-5,-7,-9 mod q and rotations)
    fq2 = "src/building_block/curves/bls12_381/fq2.rs"
    k["fq2"] = {"src": fq2 + ":166-226", "inputs": "fq_test_helper.rs:9-34; a2=Fq2(a1,b1), b2=Fq2(c1,d1)",
                "add": S(ints(fq2, 166, 175)), "sub": S(ints(fq2, 178, 187)), "mul": S(ints(fq2, 190, 199)),
                "inv_a": S(ints(fq2, 206, 211)), "inv_b": S(ints(fq2, 212, 217)), "reduce_mul": S(ints(fq2, 220, 228))}
    fq6 = "src/building_block/curves/bls12_381/fq6.rs"
    k["fq6"] = {"src": fq6 + ":190-275", "inputs": "a6=Fq6(a2,b2,c2), b6=Fq6(b2,c2,d2)",
                "add": S(ints(fq6, 190, 204)), "sub": S(ints(fq6, 205, 219)), "mul": S(ints(fq6, 220, 234)),
                "inv_a": S(ints(fq6, 239, 248)), "inv_b": S(ints(fq6, 249, 260)), "reduce_mul": S(ints(fq6, 262, 276))}
    fq12 = "src/building_block/curves/bls12_381/fq12.rs"
    k["fq12"] = {"src": fq12 + ":198-329", "inputs": "a12=Fq12(a6,b6), b12=Fq12(c6,d6); pow: 3^4=81",
                 "add": S(ints(fq12, 208, 232)), "sub": S(ints(fq12, 233, 258)), "mul": S(ints(fq12, 259, 284)),
                 "inv_a": S(ints(fq12, 289, 308)), "inv_b": S(ints(fq12, 309, 330))}

    g1 = "src/building_block/curves/bls12_381/g1_point.rs"
    v = ints(g1, 224, 230)
    k["g1_double"] = {"src": g1 + ":224-237", "x": str(v[0]), "y": str(v[1])}
    v = ints(g1, 315, 328)
    k["g1_multiples"] = {"src": g1 + ":315-333", "points": [[str(v[2 * i]), str(v[2 * i + 1])] for i in range(10)]}
    v = ints(g1, 352, 358)
    k["g1_scalar_mul"] = {"src": g1 + ":352-371 (scalar passed as an Fq element)",
                          "cases": [{"x": str(v[3 * i]), "y": str(v[3 * i + 1]), "k": str(v[3 * i + 2])} for i in range(4)]}
    k["g1_add_table"] = {"src": g1 + ":389-412", "cases": [[1, 1, 2], [1, 2, 3], [2, 2, 4], [2, 6, 8], [3, 4, 7], [5, 1, 6], [5, 2, 7], [8, 1, 9], [9, 1, 10]]}

    g2 = "src/building_block/curves/bls12_381/g2_point.rs"
    v = ints(g2, 203, 220)
    k["g2_double"] = {"src": g2 + ":199-230", "x_u1": str(v[0]), "x_u0": str(v[1]), "y_u1": str(v[2]), "y_u0": str(v[3])}
    v = ints(g2, 320, 331)
    k["g2_multiples"] = {"src": g2 + ":320-338", "order": "x1,x0,y1,y0",
                         "points": [S(v[4 * i:4 * i + 4]) for i in range(10)]}
    v = ints(g2, 357, 362)
    k["g2_scalar_mul"] = {"src": g2 + ":357-403", "cases": [{"k": str(v[0]), "x1": str(v[1]), "x0": str(v[2]), "y1": str(v[3]), "y0": str(v[4])}]}
    k["g2_add_table"] = {"src": g2 + ":421-444", "cases": [[1, 1, 2], [1, 2, 3], [2, 2, 4], [2, 6, 8], [3, 4, 7], [5, 1, 6], [5, 2, 7], [8, 1, 9], [9, 1, 10]]}

    sp = "src/building_block/curves/secp256k1/affine_point.rs"
    v = ints(sp, 193, 195)
    k["secp_double"] = {"src": sp + ":190-203", "x": str(v[0]), "y": str(v[1])}
    v = [t for _, t in lits(sp, 293, 303)]
    # each row: _n, x, y
    k["secp_multiples"] = {"src": sp + ":292-311", "points": [[v[3 * i + 1], v[3 * i + 2]] for i in range(10)], "enc": "hex"}
    v = [t for _, t in lits(sp, 332, 358)]
    k["secp_scalar_mul"] = {"src": sp + ":331-380 (k reduced into the base field before use)", "enc": "hex",
                            "cases": [{"k": v[3 * i], "x": v[3 * i + 1], "y": v[3 * i + 2]} for i in range(5)]}
    v = [t for _, t in lits(sp, 384, 399)]
    k["secp_add_large"] = {"src": sp + ":383-424", "enc": "hex (the _n fields are decimal)",
                           "p1": [v[1], v[2]], "p2": [v[4], v[5]], "p3": [v[7], v[8]]}
    k["secp_add_table"] = {"src": sp + ":402-416", "cases": [[1, 2, 3], [2, 2, 4], [2, 6, 8], [3, 4, 7], [5, 1, 6], [5, 2, 7], [8, 1, 9], [9, 1, 10]]}

    with open(OUT, "w") as f:
        json.dump(k, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    sys.exit(main())
