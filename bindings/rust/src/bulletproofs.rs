//! Bulletproofs over secp256k1 — zk/wo_trusted_setup/bulletproofs.rs:14-147.  The reference draws its challenges and blinding values
//! inside (`:42`, `:76-102`); the caller passes them here, in the order documented at zkt_bp_range_proof in include/zkt.h.
use crate::ffi;
use crate::field::{PrimeField, PrimeFieldElems, SecpFr, SecpN};
use crate::points::{AffinePoint, AffinePoints, SecpPoint};
use crate::{check_bool, init};

pub struct Bulletproofs();

/// generators gg, hh, u kept on the device with their window multiples (zkt_bp_ipa_ctx): a prover reuses them for every argument and range proof
pub struct Generators { ctx: *mut ffi::zkt_bp_ipa_ctx }
unsafe impl Send for Generators {}
impl Generators {
    pub fn new(gg: &[SecpPoint], hh: &[SecpPoint], u: &SecpPoint) -> Self {
        init();
        assert_eq!(gg.len(), hh.len());
        let (g, h) = (raw(gg), raw(hh));
        let mut ctx = std::ptr::null_mut();
        crate::check(unsafe { ffi::zkt_bp_ipa_ctx_create(gg.len(), g.as_ptr(), h.as_ptr(), &u.to_raw(), &mut ctx) });
        Generators { ctx }
    }
    /// bulletproofs.rs:19-55 over the resident generators
    #[allow(non_snake_case)]
    pub fn inner_product_argument(&self, P: &SecpPoint, a: &[SecpFr], b: &[SecpFr], xs: &[SecpFr]) -> bool {
        let (fa, fb, fx) = (SecpFr::flatten(a), SecpFr::flatten(b), SecpFr::flatten(xs));
        check_bool(unsafe { ffi::zkt_bp_inner_product_argument_ctx(self.ctx, &P.to_raw(), fa.as_ptr(), fb.as_ptr(), fx.as_ptr(), std::ptr::null_mut()) })
    }
    /// bulletproofs.rs:58-147 over the resident generators
    #[allow(non_snake_case)]
    pub fn range_proof(&self, V: &SecpPoint, aL: &[SecpFr], gamma: &SecpFr, g: &SecpPoint, h: &SecpPoint, use_inner_product_argument: bool, rnd: &[SecpFr], xs: &[SecpFr]) -> bool {
        let (fa, fr, fx) = (SecpFr::flatten(aL), SecpFr::flatten(rnd), SecpFr::flatten(xs));
        check_bool(unsafe { ffi::zkt_bp_range_proof_ctx(self.ctx, &V.to_raw(), fa.as_ptr(), gamma.limbs.as_ptr(), &g.to_raw(), &h.to_raw(), use_inner_product_argument as i32, fr.as_ptr(), fx.as_ptr(), std::ptr::null_mut()) })
    }
}
impl Drop for Generators { fn drop(&mut self) { unsafe { ffi::zkt_bp_ipa_ctx_free(self.ctx) } } }

fn raw(v: &[SecpPoint]) -> Vec<ffi::zkt_secp_affine> { v.iter().map(|p| p.to_raw()).collect() }

impl Bulletproofs {
    /// `inner_product_argument` with the reference's signature (bulletproofs.rs:19-27).  The reference draws one challenge per level from
    /// `AffinePoint::curve_group().rand_elem(true)` (:42), independently of L and R; the same draws are made here, level by level, and handed to the library.
    #[allow(non_snake_case)]
    pub fn inner_product_argument(n: &usize, gg: &AffinePoints, hh: &AffinePoints, u: &AffinePoint, P: &AffinePoint, a: &PrimeFieldElems<SecpN>, b: &PrimeFieldElems<SecpN>) -> bool {
        let f_n = PrimeField::<SecpN>::new();
        let mut levels = 0usize; let mut k = *n; while k > 1 { k /= 2; levels += 1; }
        let xs: Vec<SecpFr> = (0..levels).map(|_| f_n.rand_elem(true)).collect();
        Bulletproofs::inner_product_argument_with(*n, &gg.points, &hh.points, u, P, &a.0, &b.0, &xs)
    }
    /// `range_proof` with the reference's signature (bulletproofs.rs:58-68): alpha, sL, sR, rho, y, z, tau1, tau2, x are drawn in the reference's order
    /// (:76-102); with the inner-product argument also u = rand_point (:139) and one challenge per level.
    #[allow(non_snake_case)]
    pub fn range_proof(n: &usize, V: &AffinePoint, aL: &PrimeFieldElems<SecpN>, gamma: &SecpFr, g: &AffinePoint, h: &AffinePoint, gg: &AffinePoints, hh: &AffinePoints,
                       use_inner_product_argument: bool) -> bool {
        let f_n = PrimeField::<SecpN>::new();
        let alpha = f_n.rand_elem(true);
        let (sL, sR) = (f_n.rand_elems(n, true), f_n.rand_elems(n, true));
        let rho = f_n.rand_elem(true);
        let (y, z) = (f_n.rand_elem(true), f_n.rand_elem(true));
        let (tau1, tau2) = (f_n.rand_elem(true), f_n.rand_elem(true));
        let x = f_n.rand_elem(true);
        let mut rnd = vec![alpha, rho, y, z, tau1, tau2, x];                    // the order zkt_bp_range_proof documents (include/zkt.h)
        rnd.extend(sL.0.iter().cloned()); rnd.extend(sR.0.iter().cloned());
        let u = if use_inner_product_argument { &SecpPoint::g() * &f_n.rand_elem(true) } else { SecpPoint::AtInfinity };      // AffinePoint::rand_point(true)
        let mut levels = 0usize; let mut k = *n; while k > 1 { k /= 2; levels += 1; }
        let xs: Vec<SecpFr> = if use_inner_product_argument { (0..levels).map(|_| f_n.rand_elem(true)).collect() } else { vec![] };
        Bulletproofs::range_proof_with(*n, V, &aL.0, gamma, g, h, &gg.points, &hh.points, use_inner_product_argument, &rnd, &u, &xs)
    }
    /// bulletproofs.rs:19-55 with the challenges injected; `xs` = one challenge per level
    #[allow(non_snake_case)]
    pub fn inner_product_argument_with(n: usize, gg: &[SecpPoint], hh: &[SecpPoint], u: &SecpPoint, P: &SecpPoint, a: &[SecpFr], b: &[SecpFr], xs: &[SecpFr]) -> bool {
        init();
        assert!(gg.len() == n && hh.len() == n && a.len() == n && b.len() == n, "Tried to operate on vectors of different length");
        let (g, h) = (raw(gg), raw(hh));
        let (fa, fb, fx) = (SecpFr::flatten(a), SecpFr::flatten(b), SecpFr::flatten(xs));
        check_bool(unsafe { ffi::zkt_bp_inner_product_argument(n, g.as_ptr(), h.as_ptr(), &u.to_raw(), &P.to_raw(), fa.as_ptr(), fb.as_ptr(), fx.as_ptr(), std::ptr::null_mut()) })
    }
    /// bulletproofs.rs:58-147 with every draw injected; rnd = alpha, rho, y, z, tau1, tau2, x, sL[n], sR[n]
    #[allow(non_snake_case)]
    pub fn range_proof_with(n: usize, V: &SecpPoint, aL: &[SecpFr], gamma: &SecpFr, g: &SecpPoint, h: &SecpPoint, gg: &[SecpPoint], hh: &[SecpPoint], use_inner_product_argument: bool,
                       rnd: &[SecpFr], u: &SecpPoint, xs: &[SecpFr]) -> bool {
        init();
        assert!(gg.len() == n && hh.len() == n && aL.len() == n && rnd.len() == 7 + 2 * n);
        let (gr, hr) = (raw(gg), raw(hh));
        let (fa, fr, fx) = (SecpFr::flatten(aL), SecpFr::flatten(rnd), SecpFr::flatten(xs));
        check_bool(unsafe { ffi::zkt_bp_range_proof(n, &V.to_raw(), fa.as_ptr(), gamma.limbs.as_ptr(), &g.to_raw(), &h.to_raw(), gr.as_ptr(), hr.as_ptr(), use_inner_product_argument as i32,
                                                    fr.as_ptr(), &u.to_raw(), fx.as_ptr(), std::ptr::null_mut()) })
    }
}
