"""The Rust host side (bindings/rust/) is shipped as source — this image has no rustc — so what CAN be checked is checked here:
  * src/ffi.rs is exactly what tools/gen_rust_ffi.py generates from include/zkt.h (not stale);
  * its extern block declares every symbol the library exports (zk.exported_symbols()), with the arity of the C prototype;
  * every `ffi::zkt_*` the hand-written modules use is declared there, and every call passes that many arguments;
  * the reference's type and method names the north star asks for are present."""
import importlib, os, re, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RS = os.path.join(ROOT, "bindings", "rust", "src")
zk = importlib.import_module("zk-toolkit_amd")


def _split_top(s):
    """split on commas outside (), [], {}, <>"""
    out, depth, cur = [], 0, ""
    for i, ch in enumerate(s):
        if ch in "([{": depth += 1
        elif ch in ")]}": depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip(): out.append(cur)
    return out


def _ffi_decls():
    txt = open(os.path.join(RS, "ffi.rs")).read()
    block = txt[txt.index('extern "C" {'):]
    return {m.group(1): len(_split_top(m.group(2))) for m in re.finditer(r"pub fn (zkt_\w+)\((.*?)\)(?: -> [^;]+)?;", block)}


def _header_arity():
    txt = re.sub(r"/\*.*?\*/", " ", open(zk.HEADER).read(), flags=re.S)
    txt = re.sub(r"typedef[^;{]*\{.*?\}[^;]*;", " ", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(zkt_\w+)\s*\(([^)]*)\)\s*;", txt):
        params = m.group(2).strip()
        out[m.group(1)] = 0 if params in ("", "void") else len(params.split(","))
    return out


def test_ffi_rs_is_generated_from_the_header():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_extern_block_covers_every_export_with_matching_arity():
    decl, hdr = _ffi_decls(), _header_arity()
    exported = zk.exported_symbols()
    assert len(exported) >= 150
    missing = [s for s in exported if s not in decl]
    assert not missing, f"not declared in ffi.rs: {missing}"
    wrong = [(s, decl[s], hdr[s]) for s in exported if decl[s] != hdr[s]]
    assert not wrong, f"arity differs (symbol, ffi.rs, zkt.h): {wrong}"
    L = zk.lib()                                   # and the library really exports them
    assert all(hasattr(L, s) for s in decl)


def test_handwritten_modules_call_declared_functions_with_the_right_arity():
    decl = _ffi_decls()
    types = set(re.findall(r"pub (?:struct|type) (zkt_\w+)", open(os.path.join(RS, "ffi.rs")).read()))
    seen = set()
    for fn in sorted(os.listdir(RS)):
        if fn == "ffi.rs" or not fn.endswith(".rs"): continue
        src = re.sub(r"//.*", "", open(os.path.join(RS, fn)).read())
        for name in re.findall(r"\b(zkt_\w+)\b", src):            # also the names handed to macros without the ffi:: prefix
            assert name in decl or name in types, f"{fn}: {name} is not declared in ffi.rs"
            if name in decl: seen.add(name)
        for m in re.finditer(r"ffi::(zkt_\w+)", src):
            name = m.group(1)
            if name in types: continue
            rest = src[m.end():].lstrip()
            if not rest.startswith("("): continue           # passed as a function value, not called
            depth, j = 0, 0
            for j, ch in enumerate(rest):
                if ch == "(": depth += 1
                elif ch == ")":
                    depth -= 1
                    if depth == 0: break
            nargs = len(_split_top(rest[1:j])) if rest[1:j].strip() else 0
            assert nargs == decl[name], f"{fn}: ffi::{name} called with {nargs} arguments, declared with {decl[name]}"
    assert len(seen) >= 60, f"the shim binds only {len(seen)} entry points"


def test_reference_names_are_present():
    src = "".join(open(os.path.join(RS, f)).read() for f in os.listdir(RS) if f.endswith(".rs"))
    for needle in ("pub type Fq1", "pub struct Fq2", "pub struct Fq6", "pub struct Fq12", "pub enum G1Point", "pub enum G2Point", "pub struct GTPoint", "pub struct Pairing",
                   "pub fn tate(", "pub fn weil(", "pub fn calc_g1_g2(", "pub fn calc_g2_g1(", "pub fn eval_with_g1_hidings(", "pub fn eval_with_g2_hidings(", "pub struct CRS",
                   "pub struct Prover", "pub fn prove(", "pub struct Verifier", "pub fn verify(", "pub struct Proof", "pub fn inner_product_argument(", "pub fn range_proof(",
                   "pub fn pow(", "pub fn pow_seq(", "pub fn repeat(", "pub fn cube(", "pub fn safe_inv(", "pub fn init()"):
        assert needle in src, needle


def _norm(t):
    return re.sub(r"\s+", " ", t)


def test_reference_signatures_are_present_verbatim():
    """VERDICT r2, missing #3: the entry points of the proof systems exist with the REFERENCE's signatures (they draw their randomness where the reference
    does and call the injected forms), so that code written against the crate compiles unchanged:
      crs.rs:49-53, prover.rs:96, verifier.rs:30-35, bulletproofs.rs:19-27 and 58-68 (generic field parameters spelled out where Rust needs them)."""
    g16 = _norm(open(os.path.join(RS, "groth16.rs")).read())
    bp = _norm(open(os.path.join(RS, "bulletproofs.rs")).read())
    assert "pub fn new(f: &PrimeField<Bls12R>, prover: &Prover, pairing: &Pairing) -> Self" in g16
    assert "pub fn prove(&self, crs: &CRS) -> Proof" in g16
    assert "pub fn verify(&self, proof: &Proof, crs: &CRS, stmt_wires: &SparseVec<Bls12R>) -> bool" in g16
    assert ("pub fn inner_product_argument(n: &usize, gg: &AffinePoints, hh: &AffinePoints, u: &AffinePoint, P: &AffinePoint, a: &PrimeFieldElems<SecpN>, "
            "b: &PrimeFieldElems<SecpN>) -> bool") in bp
    assert ("pub fn range_proof(n: &usize, V: &AffinePoint, aL: &PrimeFieldElems<SecpN>, gamma: &SecpFr, g: &AffinePoint, h: &AffinePoint, gg: &AffinePoints, "
            "hh: &AffinePoints, use_inner_product_argument: bool) -> bool") in bp
    # the draws happen in the reference's order: crs.rs:59-63, prover.rs:100-101, bulletproofs.rs:76-102
    assert re.search(r"alpha: f\.rand_elem\(true\), beta: f\.rand_elem\(true\), gamma: f\.rand_elem\(true\), delta: f\.rand_elem\(true\), x: f\.rand_elem\(true\)", g16)
    assert "let (r, s) = (self.f.rand_elem(true), self.f.rand_elem(true));" in g16
    order = [bp.index(k) for k in ("let alpha = f_n.rand_elem(true)", "f_n.rand_elems(n, true), f_n.rand_elems(n, true)", "let rho = f_n.rand_elem(true)",
                                   "let (y, z) =", "let (tau1, tau2) =", "let x = f_n.rand_elem(true)")]
    assert order == sorted(order)
    # the struct the reference's CRS::new reads from carries the reference's fields (prover.rs:35-46)
    assert re.search(r"pub struct Prover \{ pub f: PrimeField<Bls12R>, pub n: usize, pub l: usize, pub m: usize, pub wires: Vec<Fr>, pub h: Vec<Fr>, pub t: Vec<Fr>, "
                     r"pub ui: Vec<Vec<Fr>>, pub vi: Vec<Vec<Fr>>, pub wi: Vec<Vec<Fr>> \}", g16)
    # Pinocchio: crs.rs:48-51, prover.rs:96, verifier.rs:31-36; draws in the reference's order (crs.rs:58-64,82; prover.rs:103-104); struct fields of crs.rs:12-39, proof.rs:8-18
    pin = _norm(open(os.path.join(RS, "pinocchio.rs")).read())
    assert "pub fn new(f: &PrimeField<Bls12R>, p: &Prover) -> Self" in pin
    assert "pub fn prove(&self, crs: &CRS) -> Proof" in pin
    assert "pub fn verify(&self, proof: &Proof, crs: &CRS, witness_io: &SparseVec<Bls12R>) -> bool" in pin
    assert re.search(r"r_v: f\.rand_elem\(true\), r_w: f\.rand_elem\(true\), alpha_v: f\.rand_elem\(true\), alpha_w: f\.rand_elem\(true\), alpha_y: f\.rand_elem\(true\), beta: f\.rand_elem\(true\), "
                     r"gamma: f\.rand_elem\(true\), s: f\.rand_elem\(true\)", pin)
    assert "let (delta_v, delta_y) = (self.f.rand_elem(true), self.f.rand_elem(true));" in pin
    for field in ("vk_mid", "g1_wk_mid", "g2_wk_mid", "yk_mid", "alpha_vk_mid", "alpha_wk_mid", "alpha_yk_mid", "si", "beta_vwy_k_mid", "one_g1", "one_g2", "alpha_v", "alpha_w", "alpha_y",
                  "gamma", "beta_gamma", "vk_io", "wk_io", "yk_io", "alpha_v_t", "alpha_y_t", "beta_t", "v_mid_s", "g1_w_mid_s", "g2_w_mid_s", "y_mid_s", "h_s", "alpha_v_mid_s",
                  "alpha_w_mid_s", "alpha_y_mid_s", "beta_vwy_mid_s"):
        assert re.search(r"pub %s: (Vec<)?G[12]Point" % field, pin), field
    src = "".join(open(os.path.join(RS, f)).read() for f in os.listdir(RS) if f.endswith(".rs"))
    for needle in ("pub struct PrimeField<", "pub fn rand_elem(&self, exclude_zero: bool)", "pub struct PrimeFieldElems<", "pub fn sum(&self)", "pub struct AffinePoints",
                   "pub type AffinePoint = SecpPoint", "pub struct SparseVec<"):
        assert needle in src, needle


def test_rust_sources_are_balanced():
    """no rustc here: at least every brace, bracket and parenthesis of the hand-written modules closes (string and comment contents skipped)"""
    for fn in sorted(os.listdir(RS)):
        if not fn.endswith(".rs"): continue
        src = open(os.path.join(RS, fn)).read()
        src = re.sub(r"//.*", "", src)
        src = re.sub(r'b?"(?:\\.|[^"\\])*"', '""', src)
        src = re.sub(r"'(?:\\.|[^'\\])'", "' '", src)
        stack = []
        for ch in src:
            if ch in "([{": stack.append(ch)
            elif ch in ")]}":
                assert stack and "([{".index(stack[-1]) == ")]}".index(ch), f"{fn}: unbalanced {ch}"
                stack.pop()
        assert not stack, f"{fn}: unclosed {stack[-3:]}"


def test_reference_module_paths_and_runtime_order_field_types():
    """The reference's call sites import by module path and pass the field as a VALUE (`PrimeField { order }`, prime_field.rs:15-33; elements `{ f: Arc<PrimeField>, e }`,
    prime_field_elem.rs:57-61).  reference_paths.rs must offer both: the paths as re-exports of the shim's types, and non-generic field types with the reference's
    constructor signatures that look the order up at run time."""
    src = open(os.path.join(ROOT, "bindings", "rust", "src", "reference_paths.rs")).read()
    lib = open(os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")).read()
    assert "pub use reference_paths::{building_block, zk};" in lib
    def has_path(path, item):
        """`pub mod a { pub mod b { ... pub use ...item` nested in that order"""
        pos = 0
        for seg in path.split("::"):
            pos = src.find("pub mod %s" % seg, pos)
            assert pos >= 0, f"{path}: module {seg} missing"
        tail = src[pos:pos + 400]
        assert re.search(r"pub (use [^;]*\b%s\b|struct %s\b)" % (item, item), tail), f"{path}::{item} not exported"
    for path, item in [("building_block::curves::bls12_381::g1_point", "G1Point"), ("building_block::curves::bls12_381::g2_point", "G2Point"),
                       ("building_block::curves::bls12_381::pairing", "Pairing"), ("building_block::curves::bls12_381::gt_point", "GTPoint"),
                       ("building_block::curves::bls12_381::fq12", "Fq12"), ("building_block::curves::bls12_381::signature", "Signer"),
                       ("building_block::curves::secp256k1::affine_points", "AffinePoints"), ("building_block::field::prime_field", "PrimeField"),
                       ("building_block::field::prime_field_elem", "PrimeFieldElem"), ("building_block::field::sparse_vec", "SparseVec"),
                       ("zk::w_trusted_setup::groth16::zktoolkit_based::prover", "Prover"), ("zk::w_trusted_setup::groth16::zktoolkit_based::verifier", "Verifier"),
                       ("zk::w_trusted_setup::groth16::zktoolkit_based::crs", "CRS"), ("zk::w_trusted_setup::pinocchio::verifier", "Verifier"),
                       ("zk::wo_trusted_setup::bulletproofs", "Bulletproofs")]:
        has_path(path, item)
    # the reference's own shapes, verbatim
    assert "pub struct PrimeField { order: BigUint }" in src                                         # prime_field.rs:16-18
    assert "pub fn new(order: &impl ToBigUint) -> Self" in src                                       # prime_field.rs:21
    assert "pub struct PrimeFieldElem { pub f: Arc<PrimeField>, pub e: BigUint }" in src             # prime_field_elem.rs:57-61
    assert "pub fn new(f: &Arc<PrimeField>, e: &impl ToBigUint) -> Self" in src                      # prime_field_elem.rs:263
    assert "pub fn rand_elem(&self, exclude_zero: bool) -> PrimeFieldElem" in src                    # prime_field.rs:73
    for needle in ("pub fn plus(&self, rhs: &impl ToBigUint) -> Self", "pub fn safe_inv(&self) -> Result<Self, String>", "pub fn pow_seq(&self, n: usize) -> Vec<Self>"):
        assert needle in src, needle
    # every re-exported item exists in the module it is re-exported from
    for mod, item in re.findall(r"pub use crate::(\w+)::\{?([\w, ]+)\}?;", src):
        body = open(os.path.join(ROOT, "bindings", "rust", "src", mod + ".rs")).read()
        for it in [x.strip() for x in item.split(",")]:
            assert re.search(r"pub (struct|enum|type|trait) %s\b" % it, body), f"{mod}::{it} does not exist"
