//! Groth16 — zk/w_trusted_setup/groth16/zktoolkit_based/{crs.rs:17-146, prover.rs:35-147, verifier.rs:19-54, proof.rs:7-11}.
//! Two forms, as in include/zkt.h: the reference's own dense-QAP interface (`CRS::new` / `Prover::prove` on ui/vi/wi coefficient
//! arrays) and the sparse-R1CS form that scales to 2^20 constraints (`ProvingKey`), which produces the SAME proof points.
//! The reference draws alpha, beta, gamma, delta, x (crs.rs:59-63) and r, s (prover.rs:100-101) from OS entropy inside; a drop-in
//! caller keeps doing that and passes the draws in here (`Trapdoor`, `r`, `s`), which is also what makes results reproducible.
use crate::ffi::{self, zkt_g1_affine, zkt_g2_affine, zkt_groth16_crs, zkt_sparse_rows};
use crate::field::{Bls12R, Fr, PrimeField, SparseVec};
use crate::pairing::{GTPoint, Pairing};
use crate::points::{G1Point, G2Point};
use crate::tower::{Fq12, Limbs};
use crate::{check, check_bool, init};

#[allow(non_snake_case)]
#[derive(Clone, Debug)]
pub struct Proof { pub A: G1Point, pub B: G2Point, pub C: G1Point } // proof.rs:7-11

#[derive(Clone, Debug)]
pub struct Trapdoor { pub alpha: Fr, pub beta: Fr, pub gamma: Fr, pub delta: Fr, pub x: Fr } // crs.rs:59-63

pub struct G1 { pub alpha: G1Point, pub beta: G1Point, pub delta: G1Point, pub xi: Vec<G1Point>, pub uvw_stmt: Vec<G1Point>, pub uvw_wit: Vec<G1Point>, pub xt_by_delta: Vec<G1Point> } // crs.rs:17-25
pub struct G2 { pub beta: G2Point, pub gamma: G2Point, pub delta: G2Point, pub xi: Vec<G2Point> } // crs.rs:27-32
pub struct GT { pub alpha_beta: GTPoint } // crs.rs:34-36
pub struct CRS { pub g1: G1, pub g2: G2, pub gt: GT, n: usize, l: usize, m: usize } // crs.rs:39-43

/// owned storage behind a zkt_groth16_crs
struct CrsBuf {
    a: Vec<zkt_g1_affine>, // alpha, beta, delta
    xi1: Vec<zkt_g1_affine>, stmt: Vec<zkt_g1_affine>, wit: Vec<zkt_g1_affine>, xt: Vec<zkt_g1_affine>,
    b: Vec<zkt_g2_affine>, // beta, gamma, delta
    xi2: Vec<zkt_g2_affine>, ab: Vec<u64>,
}
impl CrsBuf {
    fn new(n: usize, l: usize, m: usize) -> Self {
        let z1 = G1Point::zero_raw(); let z2 = G2Point::zero_raw();
        CrsBuf { a: vec![z1; 3], xi1: vec![z1; n], stmt: vec![z1; l + 1], wit: vec![z1; m - l], xt: vec![z1; n], b: vec![z2; 3], xi2: vec![z2; n], ab: vec![0u64; 72] }
    }
    fn view(&mut self, n: usize, l: usize, m: usize) -> zkt_groth16_crs {
        zkt_groth16_crs { n, l, m, g1_alpha: &mut self.a[0], g1_beta: &mut self.a[1], g1_delta: &mut self.a[2], g1_xi: self.xi1.as_mut_ptr(), g1_uvw_stmt: self.stmt.as_mut_ptr(),
                          g1_uvw_wit: self.wit.as_mut_ptr(), g1_xt_by_delta: self.xt.as_mut_ptr(), g2_beta: &mut self.b[0], g2_gamma: &mut self.b[1], g2_delta: &mut self.b[2],
                          g2_xi: self.xi2.as_mut_ptr(), gt_alpha_beta: self.ab.as_mut_ptr() }
    }
    fn into_crs(self, n: usize, l: usize, m: usize) -> CRS {
        let g1v = |v: &Vec<zkt_g1_affine>| v.iter().map(G1Point::from_raw).collect::<Vec<_>>();
        let g2v = |v: &Vec<zkt_g2_affine>| v.iter().map(G2Point::from_raw).collect::<Vec<_>>();
        CRS { g1: G1 { alpha: G1Point::from_raw(&self.a[0]), beta: G1Point::from_raw(&self.a[1]), delta: G1Point::from_raw(&self.a[2]), xi: g1v(&self.xi1), uvw_stmt: g1v(&self.stmt),
                       uvw_wit: g1v(&self.wit), xt_by_delta: g1v(&self.xt) },
              g2: G2 { beta: G2Point::from_raw(&self.b[0]), gamma: G2Point::from_raw(&self.b[1]), delta: G2Point::from_raw(&self.b[2]), xi: g2v(&self.xi2) },
              gt: GT { alpha_beta: GTPoint::new(&Fq12::read(&self.ab)) }, n, l, m }
    }
    fn from_crs(c: &CRS) -> Self {
        let g1v = |v: &Vec<G1Point>| v.iter().map(|p| p.to_raw()).collect::<Vec<_>>();
        let g2v = |v: &Vec<G2Point>| v.iter().map(|p| p.to_raw()).collect::<Vec<_>>();
        CrsBuf { a: vec![c.g1.alpha.to_raw(), c.g1.beta.to_raw(), c.g1.delta.to_raw()], xi1: g1v(&c.g1.xi), stmt: g1v(&c.g1.uvw_stmt), wit: g1v(&c.g1.uvw_wit), xt: g1v(&c.g1.xt_by_delta),
                 b: vec![c.g2.beta.to_raw(), c.g2.gamma.to_raw(), c.g2.delta.to_raw()], xi2: g2v(&c.g2.xi), ab: c.gt_limbs() }
    }
}
/// dense QAP polynomials as (m+1) x n coefficient rows, low degree first (Prover.ui/vi/wi, prover.rs:43-45)
fn dense(polys: &[Vec<Fr>], n: usize) -> Vec<u64> {
    let mut out = vec![0u64; polys.len() * n * 4];
    for (i, p) in polys.iter().enumerate() { for (k, c) in p.iter().enumerate() { out[(i * n + k) * 4..(i * n + k) * 4 + 4].copy_from_slice(&c.limbs); } }
    out
}

impl CRS {
    fn gt_limbs(&self) -> Vec<u64> { self.gt.alpha_beta.e_clone().to_vec() }
    /// `CRS::new` with the reference's signature (crs.rs:49-53): draws alpha, beta, gamma, delta, x from `f.rand_elem(true)` in that order (crs.rs:59-63)
    /// and builds sigma from the prover's QAP polynomials.  (`pairing` is what the reference uses for `e(alpha, beta)`, crs.rs:137-139: the library
    /// computes the same `Pairing::tate` value.)
    pub fn new(f: &PrimeField<Bls12R>, prover: &Prover, pairing: &Pairing) -> Self {
        let _ = pairing;
        let t = Trapdoor { alpha: f.rand_elem(true), beta: f.rand_elem(true), gamma: f.rand_elem(true), delta: f.rand_elem(true), x: f.rand_elem(true) };
        CRS::new_with_trapdoor(prover.n, prover.l, prover.m, &prover.ui, &prover.vi, &prover.wi, &t)
    }
    /// CRS::new (crs.rs:49-146) with the trapdoor injected (reproducible; what the parity tests drive)
    pub fn new_with_trapdoor(n: usize, l: usize, m: usize, ui: &[Vec<Fr>], vi: &[Vec<Fr>], wi: &[Vec<Fr>], t: &Trapdoor) -> Self {
        init();
        let mut buf = CrsBuf::new(n, l, m);
        let mut view = buf.view(n, l, m);
        let (u, v, w) = (dense(ui, n), dense(vi, n), dense(wi, n));
        check(unsafe { ffi::zkt_groth16_setup(&mut view, u.as_ptr(), v.as_ptr(), w.as_ptr(), t.alpha.limbs.as_ptr(), t.beta.limbs.as_ptr(), t.gamma.limbs.as_ptr(),
                                              t.delta.limbs.as_ptr(), t.x.limbs.as_ptr()) });
        buf.into_crs(n, l, m)
    }
}

/// prover.rs:35-46.  Polynomials are coefficient vectors, low degree first; `Prover::new` (equation parser -> gates -> R1CS -> QAP, prover.rs:48-94) is the
/// reference's symbolic front end and stays there (out of scope): a drop-in caller fills this struct from the QAP it already builds.
pub struct Prover { pub f: PrimeField<Bls12R>, pub n: usize, pub l: usize, pub m: usize, pub wires: Vec<Fr>, pub h: Vec<Fr>, pub t: Vec<Fr>, pub ui: Vec<Vec<Fr>>, pub vi: Vec<Vec<Fr>>, pub wi: Vec<Vec<Fr>> }
impl Prover {
    /// `Prover::prove` with the reference's signature (prover.rs:96): draws r, s from `self.f.rand_elem(true)` (prover.rs:100-101)
    pub fn prove(&self, crs: &CRS) -> Proof {
        let (r, s) = (self.f.rand_elem(true), self.f.rand_elem(true));
        self.prove_with(crs, &r, &s)
    }
    /// Prover::prove (prover.rs:96-147), r and s injected
    pub fn prove_with(&self, crs: &CRS, r: &Fr, s: &Fr) -> Proof {
        init();
        let mut buf = CrsBuf::from_crs(crs);
        let view = buf.view(self.n, self.l, self.m);
        let (u, v) = (dense(&self.ui, self.n), dense(&self.vi, self.n));
        let (w, h) = (Fr::flatten(&self.wires), Fr::flatten(&self.h));
        let (mut a, mut b, mut c) = (G1Point::zero_raw(), G2Point::zero_raw(), G1Point::zero_raw());
        check(unsafe { ffi::zkt_groth16_prove(&view, u.as_ptr(), v.as_ptr(), w.as_ptr(), h.as_ptr(), self.h.len(), r.limbs.as_ptr(), s.limbs.as_ptr(), &mut a, &mut b, &mut c) });
        Proof { A: G1Point::from_raw(&a), B: G2Point::from_raw(&b), C: G1Point::from_raw(&c) }
    }
}

pub struct Verifier { #[allow(dead_code)] pairing: Pairing } // verifier.rs:19-22
impl Verifier {
    pub fn new(pairing: &Pairing) -> Self { Verifier { pairing: pairing.clone() } } // :24-28
    /// `Verifier::verify` with the reference's signature (verifier.rs:30-35): the statement wires a_0..a_l as a SparseVec
    pub fn verify(&self, proof: &Proof, crs: &CRS, stmt_wires: &SparseVec<Bls12R>) -> bool { self.verify_dense(proof, crs, &stmt_wires.to_dense()) }
    /// Verifier::verify (verifier.rs:30-54): e(A,B) == alpha_beta * e(sum stmt_i uvw_stmt_i, gamma) * e(C, delta)
    pub fn verify_dense(&self, proof: &Proof, crs: &CRS, stmt_wires: &[Fr]) -> bool {
        init();
        let mut buf = CrsBuf::from_crs(crs);
        let view = buf.view(crs.n, crs.l, crs.m);
        let st = Fr::flatten(stmt_wires);
        check_bool(unsafe { ffi::zkt_groth16_verify(&view, &proof.A.to_raw(), &proof.B.to_raw(), &proof.C.to_raw(), st.as_ptr(), stmt_wires.len()) })
    }
    /// many proofs against one CRS, one proof per lane (f-2)
    pub fn verify_batch(&self, proofs: &[Proof], crs: &CRS, stmt_wires: &[Vec<Fr>]) -> Vec<bool> {
        init();
        assert_eq!(proofs.len(), stmt_wires.len());
        let mut buf = CrsBuf::from_crs(crs);
        let view = buf.view(crs.n, crs.l, crs.m);
        let a: Vec<zkt_g1_affine> = proofs.iter().map(|p| p.A.to_raw()).collect();
        let b: Vec<zkt_g2_affine> = proofs.iter().map(|p| p.B.to_raw()).collect();
        let c: Vec<zkt_g1_affine> = proofs.iter().map(|p| p.C.to_raw()).collect();
        let ns = stmt_wires.first().map(|v| v.len()).unwrap_or(0);
        let st: Vec<u64> = stmt_wires.iter().flat_map(|v| Fr::flatten(v)).collect();
        let mut ok = vec![0u32; proofs.len()];
        check(unsafe { ffi::zkt_groth16_verify_batch(&view, a.as_ptr(), b.as_ptr(), c.as_ptr(), st.as_ptr(), ns, proofs.len(), ok.as_mut_ptr()) });
        ok.iter().map(|v| *v == 1).collect()
    }
}

/// One sparse R1CS matrix, a row per constraint (R1CS.constraints; Constraint{a,b,c}, constraint.rs:5-9): CSR over wire indices
pub struct SparseRows { pub rowptr: Vec<u64>, pub col: Vec<u32>, pub val: Vec<Fr> }
/// the device-resident proving key of the sparse-R1CS form (Lagrange-basis CRS on the reference's domain {1..n})
pub struct ProvingKey { pk: *mut ffi::zkt_groth16_pk }
unsafe impl Send for ProvingKey {}
impl ProvingKey {
    /// CRS::new on the R1CS itself: returns the key and the verifying part of the CRS (xi / xt_by_delta are not produced)
    #[allow(non_snake_case)]
    pub fn setup(n: usize, l: usize, m: usize, A: &SparseRows, B: &SparseRows, C: &SparseRows, t: &Trapdoor) -> (ProvingKey, CRS) {
        init();
        let vals: Vec<Vec<u64>> = [A, B, C].iter().map(|r| Fr::flatten(&r.val)).collect();
        let rows: Vec<zkt_sparse_rows> = [A, B, C].iter().zip(vals.iter()).map(|(r, v)| zkt_sparse_rows { rowptr: r.rowptr.as_ptr(), col: r.col.as_ptr(), val: v.as_ptr() }).collect();
        let mut buf = CrsBuf::new(0, l, m);
        let mut view = buf.view(n, l, m);
        view.g1_xi = std::ptr::null_mut(); view.g1_xt_by_delta = std::ptr::null_mut(); view.g2_xi = std::ptr::null_mut();
        let mut pk = std::ptr::null_mut();
        check(unsafe { ffi::zkt_groth16_setup_r1cs(n, l, m, &rows[0], &rows[1], &rows[2], t.alpha.limbs.as_ptr(), t.beta.limbs.as_ptr(), t.gamma.limbs.as_ptr(), t.delta.limbs.as_ptr(),
                                                   t.x.limbs.as_ptr(), &mut view, &mut pk) });
        (ProvingKey { pk }, buf.into_crs(n, l, m))
    }
    /// Prover::prove (prover.rs:96-147) from the wires a_0..a_m
    pub fn prove(&self, wires: &[Fr], r: &Fr, s: &Fr) -> Proof {
        let w = Fr::flatten(wires);
        let (mut a, mut b, mut c) = (G1Point::zero_raw(), G2Point::zero_raw(), G1Point::zero_raw());
        check(unsafe { ffi::zkt_groth16_prove_r1cs(self.pk, w.as_ptr(), r.limbs.as_ptr(), s.limbs.as_ptr(), &mut a, &mut b, &mut c) });
        Proof { A: G1Point::from_raw(&a), B: G2Point::from_raw(&b), C: G1Point::from_raw(&c) }
    }
    pub fn raw(&self) -> *mut ffi::zkt_groth16_pk { self.pk }
}
impl Drop for ProvingKey { fn drop(&mut self) { unsafe { ffi::zkt_groth16_pk_free(self.pk) } } }
