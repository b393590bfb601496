//! Pinocchio (protocol 2 of eprint 2013/279) — zk/w_trusted_setup/pinocchio/{crs.rs:12-161, prover.rs:37-170, verifier.rs:18-85, proof.rs:8-18, witness.rs:6-28}.
//! `CRS::new(f, p)`, `Prover::prove(&self, crs)` and `Verifier::verify(&self, proof, crs, witness_io)` keep the reference's signatures and draw their
//! randomness where the reference does (crs.rs:58-64,82; prover.rs:103-104); the `*_with` forms take the draws as arguments (what the parity tests drive).
//! `Prover::new` (equation parser -> gates -> R1CS -> QAP, prover.rs:50-94) is the reference's symbolic front end and stays there (out of
//! scope): a drop-in caller fills `Prover` from the QAP it already builds.  `ResidentProver` keeps the evaluation key in HBM across proofs.
use crate::ffi::{self, zkt_g1_affine, zkt_g2_affine, zkt_pinocchio_crs, zkt_pinocchio_proof};
use crate::field::{Bls12R, Fr, PrimeField, SparseVec};
use crate::pairing::Pairing;
use crate::points::{G1Point, G2Point};
use crate::{check, check_bool, init};

pub struct EvaluationKeys { pub vk_mid: Vec<G1Point>, pub g1_wk_mid: Vec<G1Point>, pub g2_wk_mid: Vec<G2Point>, pub yk_mid: Vec<G1Point>, pub alpha_vk_mid: Vec<G1Point>,
                            pub alpha_wk_mid: Vec<G1Point>, pub alpha_yk_mid: Vec<G1Point>, pub si: Vec<G2Point>, pub beta_vwy_k_mid: Vec<G1Point> } // crs.rs:12-22
pub struct VerificationKeys { pub one_g1: G1Point, pub one_g2: G2Point, pub alpha_v: G2Point, pub alpha_w: G1Point, pub alpha_y: G2Point, pub gamma: G2Point, pub beta_gamma: G2Point,
                              pub t: G1Point, pub vk_io: Vec<G1Point>, pub wk_io: Vec<G2Point>, pub yk_io: Vec<G1Point>, pub alpha_v_t: G1Point, pub alpha_y_t: G1Point, pub beta_t: G1Point } // crs.rs:24-39
pub struct CRS { pub ek: EvaluationKeys, pub vk: VerificationKeys, n: usize, max_degree: usize } // crs.rs:41-44
#[derive(Clone, Debug)]
pub struct Proof { pub v_mid_s: G1Point, pub g1_w_mid_s: G1Point, pub g2_w_mid_s: G2Point, pub y_mid_s: G1Point, pub h_s: G2Point, pub alpha_v_mid_s: G1Point,
                   pub alpha_w_mid_s: G1Point, pub alpha_y_mid_s: G1Point, pub beta_vwy_mid_s: G1Point } // proof.rs:8-18
/// witness.rs:6-28: every wire value (the constant one first), `mid_beg` = index of the first mid wire
pub struct Witness { pub sv: SparseVec<Bls12R>, pub mid_beg: usize }
impl Witness {
    pub fn io(&self) -> Vec<Fr> { (0..self.mid_beg).map(|i| self.sv.get(i)).collect() }                  // :21-24
    pub fn mid(&self) -> Vec<Fr> { (self.mid_beg..self.sv.size).map(|i| self.sv.get(i)).collect() }      // :26-28
}
/// the eight draws of CRS::new in the reference's order (crs.rs:58-64, 82)
#[derive(Clone, Debug)]
pub struct Toxic { pub r_v: Fr, pub r_w: Fr, pub alpha_v: Fr, pub alpha_w: Fr, pub alpha_y: Fr, pub beta: Fr, pub gamma: Fr, pub s: Fr }

/// owned storage behind a zkt_pinocchio_crs
struct CrsBuf { g1: Vec<Vec<zkt_g1_affine>>, g2: Vec<Vec<zkt_g2_affine>> }
impl CrsBuf {
    // g1: 0 vk_mid 1 g1_wk_mid 2 yk_mid 3 alpha_vk_mid 4 alpha_wk_mid 5 alpha_yk_mid 6 beta_vwy_k_mid 7 one_g1 8 alpha_w 9 t 10 vk_io 11 yk_io 12 alpha_v_t 13 alpha_y_t 14 beta_t
    // g2: 0 g2_wk_mid 1 si 2 one_g2 3 alpha_v 4 alpha_y 5 gamma 6 beta_gamma 7 wk_io
    fn new(n_io: usize, n_mid: usize, max_degree: usize) -> Self {
        let (z1, z2) = (G1Point::zero_raw(), G2Point::zero_raw());
        let (m, io) = (n_mid.max(1), n_io.max(1));
        let l1 = [m, m, m, m, m, m, m, 1, 1, 1, io, io, 1, 1, 1];
        let l2 = [m, max_degree.max(1), 1, 1, 1, 1, 1, io];
        CrsBuf { g1: l1.iter().map(|k| vec![z1; *k]).collect(), g2: l2.iter().map(|k| vec![z2; *k]).collect() }
    }
    fn view(&mut self, n: usize, n_io: usize, n_mid: usize, max_degree: usize) -> zkt_pinocchio_crs {
        let p1: Vec<*mut zkt_g1_affine> = self.g1.iter_mut().map(|v| v.as_mut_ptr()).collect();
        let p2: Vec<*mut zkt_g2_affine> = self.g2.iter_mut().map(|v| v.as_mut_ptr()).collect();
        zkt_pinocchio_crs { n, n_io, n_mid, max_degree, vk_mid: p1[0], g1_wk_mid: p1[1], g2_wk_mid: p2[0], yk_mid: p1[2], alpha_vk_mid: p1[3], alpha_wk_mid: p1[4], alpha_yk_mid: p1[5],
                            si: p2[1], beta_vwy_k_mid: p1[6], one_g1: p1[7], one_g2: p2[2], alpha_v: p2[3], alpha_w: p1[8], alpha_y: p2[4], gamma: p2[5], beta_gamma: p2[6], t: p1[9],
                            vk_io: p1[10], wk_io: p2[7], yk_io: p1[11], alpha_v_t: p1[12], alpha_y_t: p1[13], beta_t: p1[14] }
    }
    fn into_crs(self, n: usize, n_io: usize, n_mid: usize, max_degree: usize) -> CRS {
        let v1 = |k: usize, len: usize| self.g1[k][..len].iter().map(G1Point::from_raw).collect::<Vec<_>>();
        let v2 = |k: usize, len: usize| self.g2[k][..len].iter().map(G2Point::from_raw).collect::<Vec<_>>();
        let (s1, s2) = (|k: usize| G1Point::from_raw(&self.g1[k][0]), |k: usize| G2Point::from_raw(&self.g2[k][0]));
        CRS { ek: EvaluationKeys { vk_mid: v1(0, n_mid), g1_wk_mid: v1(1, n_mid), g2_wk_mid: v2(0, n_mid), yk_mid: v1(2, n_mid), alpha_vk_mid: v1(3, n_mid), alpha_wk_mid: v1(4, n_mid),
                                   alpha_yk_mid: v1(5, n_mid), si: v2(1, max_degree), beta_vwy_k_mid: v1(6, n_mid) },
              vk: VerificationKeys { one_g1: s1(7), one_g2: s2(2), alpha_v: s2(3), alpha_w: s1(8), alpha_y: s2(4), gamma: s2(5), beta_gamma: s2(6), t: s1(9), vk_io: v1(10, n_io),
                                     wk_io: v2(7, n_io), yk_io: v1(11, n_io), alpha_v_t: s1(12), alpha_y_t: s1(13), beta_t: s1(14) },
              n, max_degree }
    }
    fn from_crs(c: &CRS) -> Self {
        let r1 = |v: &Vec<G1Point>| { let mut o: Vec<zkt_g1_affine> = v.iter().map(|p| p.to_raw()).collect(); if o.is_empty() { o.push(G1Point::zero_raw()); } o };
        let r2 = |v: &Vec<G2Point>| { let mut o: Vec<zkt_g2_affine> = v.iter().map(|p| p.to_raw()).collect(); if o.is_empty() { o.push(G2Point::zero_raw()); } o };
        let (e, k) = (&c.ek, &c.vk);
        CrsBuf { g1: vec![r1(&e.vk_mid), r1(&e.g1_wk_mid), r1(&e.yk_mid), r1(&e.alpha_vk_mid), r1(&e.alpha_wk_mid), r1(&e.alpha_yk_mid), r1(&e.beta_vwy_k_mid), vec![k.one_g1.to_raw()],
                          vec![k.alpha_w.to_raw()], vec![k.t.to_raw()], r1(&k.vk_io), r1(&k.yk_io), vec![k.alpha_v_t.to_raw()], vec![k.alpha_y_t.to_raw()], vec![k.beta_t.to_raw()]],
                 g2: vec![r2(&e.g2_wk_mid), r2(&e.si), vec![k.one_g2.to_raw()], vec![k.alpha_v.to_raw()], vec![k.alpha_y.to_raw()], vec![k.gamma.to_raw()], vec![k.beta_gamma.to_raw()], r2(&k.wk_io)] }
    }
}
impl CRS {
    fn n_io(&self) -> usize { self.vk.vk_io.len() }
    fn n_mid(&self) -> usize { self.ek.vk_mid.len() }
    /// `CRS::new` with the reference's signature (crs.rs:48-51): r_v, r_w, alpha_v, alpha_w, alpha_y, beta, gamma from `f.rand_elem(true)` in that order (crs.rs:58-64), then s (crs.rs:82)
    pub fn new(f: &PrimeField<Bls12R>, p: &Prover) -> Self {
        let t = Toxic { r_v: f.rand_elem(true), r_w: f.rand_elem(true), alpha_v: f.rand_elem(true), alpha_w: f.rand_elem(true), alpha_y: f.rand_elem(true), beta: f.rand_elem(true),
                        gamma: f.rand_elem(true), s: f.rand_elem(true) };
        CRS::new_with(p, &t)
    }
    /// CRS::new (crs.rs:48-161) with the draws injected
    pub fn new_with(p: &Prover, t: &Toxic) -> Self {
        init();
        let (n, n_io) = (p.num_constraints, p.witness.mid_beg);
        let n_mid = p.witness.sv.size - n_io;
        let mut buf = CrsBuf::new(n_io, n_mid, p.max_degree);
        let mut view = buf.view(n, n_io, n_mid, p.max_degree);
        let dense = |polys: &Vec<Vec<Fr>>| { let mut o = vec![0u64; polys.len() * n * 4]; for (i, q) in polys.iter().enumerate() { for (k, c) in q.iter().enumerate() { o[(i * n + k) * 4..(i * n + k) * 4 + 4].copy_from_slice(&c.limbs); } } o };
        let (v, w, y) = (dense(&p.vi), dense(&p.wi), dense(&p.yi));
        let rnd = Fr::flatten(&[t.r_v.clone(), t.r_w.clone(), t.alpha_v.clone(), t.alpha_w.clone(), t.alpha_y.clone(), t.beta.clone(), t.gamma.clone(), t.s.clone()]);
        check(unsafe { ffi::zkt_pinocchio_setup(&mut view, v.as_ptr(), w.as_ptr(), y.as_ptr(), rnd.as_ptr()) });
        buf.into_crs(n, n_io, n_mid, p.max_degree)
    }
}

/// prover.rs:37-47.  Polynomials are coefficient vectors, low degree first; `h` = the coefficients of p / t (prover.rs:143-146), which the reference's Prover holds as `p` and `t`.
pub struct Prover { pub f: PrimeField<Bls12R>, pub max_degree: usize, pub num_constraints: usize, pub witness: Witness, pub h: Vec<Fr>, pub vi: Vec<Vec<Fr>>, pub wi: Vec<Vec<Fr>>, pub yi: Vec<Vec<Fr>> }
fn proof_out() -> (Vec<zkt_g1_affine>, Vec<zkt_g2_affine>) { (vec![G1Point::zero_raw(); 7], vec![G2Point::zero_raw(); 2]) }
fn proof_view(g1: &mut Vec<zkt_g1_affine>, g2: &mut Vec<zkt_g2_affine>) -> zkt_pinocchio_proof {
    let p1 = g1.as_mut_ptr(); let p2 = g2.as_mut_ptr();
    unsafe { zkt_pinocchio_proof { v_mid_s: p1, g1_w_mid_s: p1.add(1), g2_w_mid_s: p2, y_mid_s: p1.add(2), h_s: p2.add(1), alpha_v_mid_s: p1.add(3), alpha_w_mid_s: p1.add(4),
                                   alpha_y_mid_s: p1.add(5), beta_vwy_mid_s: p1.add(6) } }
}
fn proof_from(g1: &Vec<zkt_g1_affine>, g2: &Vec<zkt_g2_affine>) -> Proof {
    Proof { v_mid_s: G1Point::from_raw(&g1[0]), g1_w_mid_s: G1Point::from_raw(&g1[1]), g2_w_mid_s: G2Point::from_raw(&g2[0]), y_mid_s: G1Point::from_raw(&g1[2]), h_s: G2Point::from_raw(&g2[1]),
            alpha_v_mid_s: G1Point::from_raw(&g1[3]), alpha_w_mid_s: G1Point::from_raw(&g1[4]), alpha_y_mid_s: G1Point::from_raw(&g1[5]), beta_vwy_mid_s: G1Point::from_raw(&g1[6]) }
}
impl Prover {
    fn wires(&self) -> Vec<u64> { Fr::flatten(&self.witness.sv.to_dense()) }
    /// `Prover::prove` with the reference's signature (prover.rs:96): delta_v, delta_y from `self.f.rand_elem(true)` (prover.rs:103-104)
    pub fn prove(&self, crs: &CRS) -> Proof {
        let (delta_v, delta_y) = (self.f.rand_elem(true), self.f.rand_elem(true));
        self.prove_with(crs, &delta_v, &delta_y)
    }
    /// Prover::prove (prover.rs:96-170), delta_v and delta_y injected
    pub fn prove_with(&self, crs: &CRS, delta_v: &Fr, delta_y: &Fr) -> Proof {
        init();
        let mut buf = CrsBuf::from_crs(crs);
        let view = buf.view(crs.n, crs.n_io(), crs.n_mid(), crs.max_degree);
        let (w, h) = (self.wires(), Fr::flatten(&self.h));
        let (mut g1, mut g2) = proof_out();
        let mut pv = proof_view(&mut g1, &mut g2);
        check(unsafe { ffi::zkt_pinocchio_prove(&view, w.as_ptr(), h.as_ptr(), self.h.len(), delta_v.limbs.as_ptr(), delta_y.limbs.as_ptr(), &mut pv) });
        proof_from(&g1, &g2)
    }
}
/// the prover of a deployment that proves many statements against one CRS: the evaluation key stays in HBM (zkt_pinocchio_pk)
pub struct ResidentProver { pk: *mut ffi::zkt_pinocchio_pk }
unsafe impl Send for ResidentProver {}
impl ResidentProver {
    pub fn new(crs: &CRS) -> Self {
        init();
        let mut buf = CrsBuf::from_crs(crs);
        let view = buf.view(crs.n, crs.n_io(), crs.n_mid(), crs.max_degree);
        let mut pk = std::ptr::null_mut();
        check(unsafe { ffi::zkt_pinocchio_pk_create(&view, &mut pk) });
        ResidentProver { pk }
    }
    /// the nine points of `Prover::prove_with` for the same arguments
    pub fn prove_with(&self, p: &Prover, delta_v: &Fr, delta_y: &Fr) -> Proof {
        let (w, h) = (p.wires(), Fr::flatten(&p.h));
        let (mut g1, mut g2) = proof_out();
        let mut pv = proof_view(&mut g1, &mut g2);
        check(unsafe { ffi::zkt_pinocchio_prove_resident(self.pk, w.as_ptr(), h.as_ptr(), p.h.len(), delta_v.limbs.as_ptr(), delta_y.limbs.as_ptr(), &mut pv) });
        proof_from(&g1, &g2)
    }
}
impl Drop for ResidentProver { fn drop(&mut self) { unsafe { ffi::zkt_pinocchio_pk_free(self.pk) } } }

pub struct Verifier { #[allow(dead_code)] pairing: Pairing } // verifier.rs:18-20
impl Verifier {
    pub fn new() -> Self { Verifier { pairing: Pairing::new() } } // :23-29
    /// `Verifier::verify` with the reference's signature (verifier.rs:31-36): the five equalities in the reference's order, a rejection by an earlier one wins over a panic of a later one
    pub fn verify(&self, proof: &Proof, crs: &CRS, witness_io: &SparseVec<Bls12R>) -> bool {
        init();
        let mut buf = CrsBuf::from_crs(crs);
        let view = buf.view(crs.n, crs.n_io(), crs.n_mid(), crs.max_degree);
        let mut g1 = vec![proof.v_mid_s.to_raw(), proof.g1_w_mid_s.to_raw(), proof.y_mid_s.to_raw(), proof.alpha_v_mid_s.to_raw(), proof.alpha_w_mid_s.to_raw(), proof.alpha_y_mid_s.to_raw(),
                          proof.beta_vwy_mid_s.to_raw()];
        let mut g2 = vec![proof.g2_w_mid_s.to_raw(), proof.h_s.to_raw()];
        let pv = proof_view(&mut g1, &mut g2);
        let io = Fr::flatten(&witness_io.to_dense());
        check_bool(unsafe { ffi::zkt_pinocchio_verify(&view, &pv, io.as_ptr()) })
    }
}
