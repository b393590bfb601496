//! Bulletproofs over secp256k1 — zk/wo_trusted_setup/bulletproofs.rs:14-147.  The reference draws its challenges and blinding values
//! inside (`:42`, `:76-102`); the caller passes them here, in the order documented at zkt_bp_range_proof in include/zkt.h.
use crate::ffi;
use crate::field::SecpFr;
use crate::points::SecpPoint;
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
    /// bulletproofs.rs:19-55; `xs` = one challenge per level
    #[allow(non_snake_case)]
    pub fn inner_product_argument(n: usize, gg: &[SecpPoint], hh: &[SecpPoint], u: &SecpPoint, P: &SecpPoint, a: &[SecpFr], b: &[SecpFr], xs: &[SecpFr]) -> bool {
        init();
        assert!(gg.len() == n && hh.len() == n && a.len() == n && b.len() == n, "Tried to operate on vectors of different length");
        let (g, h) = (raw(gg), raw(hh));
        let (fa, fb, fx) = (SecpFr::flatten(a), SecpFr::flatten(b), SecpFr::flatten(xs));
        check_bool(unsafe { ffi::zkt_bp_inner_product_argument(n, g.as_ptr(), h.as_ptr(), &u.to_raw(), &P.to_raw(), fa.as_ptr(), fb.as_ptr(), fx.as_ptr(), std::ptr::null_mut()) })
    }
    /// bulletproofs.rs:58-147; rnd = alpha, rho, y, z, tau1, tau2, x, sL[n], sR[n]
    #[allow(non_snake_case)]
    pub fn range_proof(n: usize, V: &SecpPoint, aL: &[SecpFr], gamma: &SecpFr, g: &SecpPoint, h: &SecpPoint, gg: &[SecpPoint], hh: &[SecpPoint], use_inner_product_argument: bool,
                       rnd: &[SecpFr], u: &SecpPoint, xs: &[SecpFr]) -> bool {
        init();
        assert!(gg.len() == n && hh.len() == n && aL.len() == n && rnd.len() == 7 + 2 * n);
        let (gr, hr) = (raw(gg), raw(hh));
        let (fa, fr, fx) = (SecpFr::flatten(aL), SecpFr::flatten(rnd), SecpFr::flatten(xs));
        check_bool(unsafe { ffi::zkt_bp_range_proof(n, &V.to_raw(), fa.as_ptr(), gamma.limbs.as_ptr(), &g.to_raw(), &h.to_raw(), gr.as_ptr(), hr.as_ptr(), use_inner_product_argument as i32,
                                                    fr.as_ptr(), &u.to_raw(), fx.as_ptr(), std::ptr::null_mut()) })
    }
}
