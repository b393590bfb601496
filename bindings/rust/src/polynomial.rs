//! Polynomial::eval_with_g1_hidings / eval_with_g2_hidings (polynomial.rs:271-293): sum_i coeffs[i] * powers[i] — the MSM.
//! Resident base sets (`G1Bases`, `G2Bases`) are the analogue of a CRS that is uploaded once and reused by every proof.
use crate::ffi;
use crate::field::Fr;
use crate::points::{G1Point, G2Point};
use crate::{check, init};

#[derive(Clone, Debug)]
pub struct Polynomial { pub coeffs: Vec<Fr> } // polynomial.rs: coefficients, low degree first

impl Polynomial {
    pub fn new(coeffs: &[Fr]) -> Self { Polynomial { coeffs: coeffs.to_vec() } }
    /// polynomial.rs:271-281 — panics if there are fewer powers than coefficients, as the reference's index does (:277-279)
    #[allow(non_snake_case)]
    pub fn eval_with_g1_hidings(&self, powers: &[G1Point]) -> G1Point {
        init();
        let n = self.coeffs.len();
        assert!(powers.len() >= n, "index out of bounds: the len is {} but the index is {}", powers.len(), powers.len());
        let p: Vec<ffi::zkt_g1_affine> = powers[..n].iter().map(|x| x.to_raw()).collect();
        let k = Fr::flatten(&self.coeffs);
        let mut out = G1Point::zero_raw();
        check(unsafe { ffi::zkt_g1_msm(p.as_ptr(), k.as_ptr(), n, &mut out) });
        G1Point::from_raw(&out)
    }
    /// polynomial.rs:283-293
    #[allow(non_snake_case)]
    pub fn eval_with_g2_hidings(&self, powers: &[G2Point]) -> G2Point {
        init();
        let n = self.coeffs.len();
        assert!(powers.len() >= n, "index out of bounds: the len is {} but the index is {}", powers.len(), powers.len());
        let p: Vec<ffi::zkt_g2_affine> = powers[..n].iter().map(|x| x.to_raw()).collect();
        let k = Fr::flatten(&self.coeffs);
        let mut out = G2Point::zero_raw();
        check(unsafe { ffi::zkt_g2_msm(p.as_ptr(), k.as_ptr(), n, &mut out) });
        G2Point::from_raw(&out)
    }
}

/// powers kept on the device with their window multiples (zkt_g1_bases): many evaluations against one CRS
pub struct G1Bases { h: *mut ffi::zkt_g1_bases, n: usize }
unsafe impl Send for G1Bases {}
impl G1Bases {
    pub fn upload(powers: &[G1Point]) -> Self {
        init();
        let p: Vec<ffi::zkt_g1_affine> = powers.iter().map(|x| x.to_raw()).collect();
        let mut h = std::ptr::null_mut();
        check(unsafe { ffi::zkt_g1_bases_upload(p.as_ptr(), p.len(), &mut h) });
        G1Bases { h, n: p.len() }
    }
    pub fn len(&self) -> usize { self.n }
    /// `dev_scalars`: n x 4 limbs already in HBM (a hipMalloc'ed buffer of the host application), on `stream`
    pub unsafe fn eval_dev(&self, dev_scalars: *const u64, stream: *mut std::os::raw::c_void) -> G1Point {
        let mut out = G1Point::zero_raw();
        check(ffi::zkt_g1_msm_dev(self.h, dev_scalars, self.n, stream, &mut out, std::ptr::null_mut()));
        G1Point::from_raw(&out)
    }
    pub fn raw(&self) -> *mut ffi::zkt_g1_bases { self.h }
}
impl Drop for G1Bases { fn drop(&mut self) { unsafe { ffi::zkt_g1_bases_free(self.h) } } }

pub struct G2Bases { h: *mut ffi::zkt_g2_bases, n: usize }
unsafe impl Send for G2Bases {}
impl G2Bases {
    pub fn upload(powers: &[G2Point]) -> Self {
        init();
        let p: Vec<ffi::zkt_g2_affine> = powers.iter().map(|x| x.to_raw()).collect();
        let mut h = std::ptr::null_mut();
        check(unsafe { ffi::zkt_g2_bases_upload(p.as_ptr(), p.len(), &mut h) });
        G2Bases { h, n: p.len() }
    }
    pub fn len(&self) -> usize { self.n }
    pub unsafe fn eval_dev(&self, dev_scalars: *const u64, stream: *mut std::os::raw::c_void) -> G2Point {
        let mut out = G2Point::zero_raw();
        check(ffi::zkt_g2_msm_dev(self.h, dev_scalars, self.n, stream, &mut out, std::ptr::null_mut()));
        G2Point::from_raw(&out)
    }
    pub fn raw(&self) -> *mut ffi::zkt_g2_bases { self.h }
}
impl Drop for G2Bases { fn drop(&mut self) { unsafe { ffi::zkt_g2_bases_free(self.h) } } }
