//! G1Point / G2Point (g1_point.rs:32-195, g2_point.rs:30-164) and secp256k1's AffinePoint (secp256k1/affine_point.rs:23-147):
//! `enum { Rational { x, y }, AtInfinity }`, affine addition with the reference's case order (curves/macros.rs:34-163) and scalar
//! multiplication by ANY PrimeFieldElem — the stored integer is used as-is, not reduced mod r (macros.rs:1-32).
use crate::field::{Fq1, FieldSpec, PrimeFieldElem, SecpFq};
use crate::ffi::{self, zkt_g1_affine, zkt_g2_affine, zkt_secp_affine};
use crate::tower::{Fq2, Limbs};
use crate::{check, init};
use std::ops::{Add, Mul, Neg};

#[derive(Clone, Debug, PartialEq, Eq)]
pub enum G1Point { Rational { x: Fq1, y: Fq1 }, AtInfinity }
#[derive(Clone, Debug, PartialEq, Eq)]
pub enum G2Point { Rational { x: Fq2, y: Fq2 }, AtInfinity }
#[derive(Clone, Debug, PartialEq, Eq)]
pub enum SecpPoint { Rational { x: SecpFq, y: SecpFq }, AtInfinity }

fn arr<const N: usize>(l: &[u64]) -> [u64; N] { let mut a = [0u64; N]; a.copy_from_slice(&l[..N]); a }

impl G1Point {
    pub fn new(x: &Fq1, y: &Fq1) -> Self { G1Point::Rational { x: x.clone(), y: y.clone() } } // g1_point.rs:50-55 (no curve check, as there)
    pub fn g() -> Self { init(); let mut p = Self::zero_raw(); unsafe { ffi::zkt_g1_generator(&mut p) }; Self::from_raw(&p) } // :57-59
    pub fn zero() -> Self { G1Point::AtInfinity }
    pub fn is_zero(&self) -> bool { matches!(self, G1Point::AtInfinity) }
    pub fn is_rational_point(&self) -> bool { // g1_point.rs:97-113
        init();
        let (p, mut ok) = (self.to_raw(), 0u32);
        check(unsafe { ffi::zkt_g1_is_on_curve_batch(&p, &mut ok, 1) });
        ok == 1
    }
    pub(crate) fn zero_raw() -> zkt_g1_affine { zkt_g1_affine { x: [0; 6], y: [0; 6], is_infinity: 1, _pad: 0 } }
    pub fn to_raw(&self) -> zkt_g1_affine {
        match self {
            G1Point::AtInfinity => Self::zero_raw(),
            G1Point::Rational { x, y } => zkt_g1_affine { x: arr(&x.limbs), y: arr(&y.limbs), is_infinity: 0, _pad: 0 },
        }
    }
    pub fn from_raw(p: &zkt_g1_affine) -> Self {
        if p.is_infinity != 0 { G1Point::AtInfinity } else { G1Point::Rational { x: Fq1::read(&p.x), y: Fq1::read(&p.y) } }
    }
    /// out[i] = scalars[i] * points[i], one launch (impl_scalar_mul_point!, macros.rs:1-32)
    pub fn mul_batch<F: FieldSpec>(points: &[G1Point], scalars: &[PrimeFieldElem<F>]) -> Vec<G1Point> {
        init();
        assert_eq!(points.len(), scalars.len());
        let p: Vec<zkt_g1_affine> = points.iter().map(|x| x.to_raw()).collect();
        let k = PrimeFieldElem::<F>::flatten(scalars);
        let mut out = vec![Self::zero_raw(); p.len()];
        check(unsafe { ffi::zkt_g1_mul_batch(p.as_ptr(), k.as_ptr(), F::LIMBS as i32, out.as_mut_ptr(), p.len()) });
        out.iter().map(Self::from_raw).collect()
    }
}
impl G2Point {
    pub fn new(x: &Fq2, y: &Fq2) -> Self { G2Point::Rational { x: x.clone(), y: y.clone() } }
    pub fn g() -> Self { init(); let mut p = Self::zero_raw(); unsafe { ffi::zkt_g2_generator(&mut p) }; Self::from_raw(&p) } // g2_point.rs:56-58
    pub fn zero() -> Self { G2Point::AtInfinity }
    pub fn is_zero(&self) -> bool { matches!(self, G2Point::AtInfinity) }
    pub(crate) fn zero_raw() -> zkt_g2_affine { zkt_g2_affine { x: [0; 12], y: [0; 12], is_infinity: 1, _pad: 0 } }
    pub fn to_raw(&self) -> zkt_g2_affine {
        match self {
            G2Point::AtInfinity => Self::zero_raw(),
            G2Point::Rational { x, y } => zkt_g2_affine { x: arr(&x.to_vec()), y: arr(&y.to_vec()), is_infinity: 0, _pad: 0 },
        }
    }
    pub fn from_raw(p: &zkt_g2_affine) -> Self {
        if p.is_infinity != 0 { G2Point::AtInfinity } else { G2Point::Rational { x: Fq2::read(&p.x), y: Fq2::read(&p.y) } }
    }
    pub fn mul_batch<F: FieldSpec>(points: &[G2Point], scalars: &[PrimeFieldElem<F>]) -> Vec<G2Point> {
        init();
        assert_eq!(points.len(), scalars.len());
        let p: Vec<zkt_g2_affine> = points.iter().map(|x| x.to_raw()).collect();
        let k = PrimeFieldElem::<F>::flatten(scalars);
        let mut out = vec![Self::zero_raw(); p.len()];
        check(unsafe { ffi::zkt_g2_mul_batch(p.as_ptr(), k.as_ptr(), F::LIMBS as i32, out.as_mut_ptr(), p.len()) });
        out.iter().map(Self::from_raw).collect()
    }
}
impl SecpPoint {
    pub fn g() -> Self { init(); let mut p = Self::zero_raw(); unsafe { ffi::zkt_secp_generator(&mut p) }; Self::from_raw(&p) } // affine_point.rs:40-60
    pub(crate) fn zero_raw() -> zkt_secp_affine { zkt_secp_affine { x: [0; 4], y: [0; 4], is_infinity: 1, _pad: 0 } }
    pub fn to_raw(&self) -> zkt_secp_affine {
        match self {
            SecpPoint::AtInfinity => Self::zero_raw(),
            SecpPoint::Rational { x, y } => zkt_secp_affine { x: arr(&x.limbs), y: arr(&y.limbs), is_infinity: 0, _pad: 0 },
        }
    }
    pub fn from_raw(p: &zkt_secp_affine) -> Self {
        if p.is_infinity != 0 { SecpPoint::AtInfinity }
        else { SecpPoint::Rational { x: PrimeFieldElem::from_canonical_limbs(&p.x), y: PrimeFieldElem::from_canonical_limbs(&p.y) } }
    }
}

macro_rules! point_ops {
    ($t:ident, $raw:ident, $add:ident, $neg:ident, $mul:ident) => {
        impl<'a> Add<&'a $t> for &'a $t { // impl_affine_add! (macros.rs:34-163)
            type Output = $t;
            fn add(self, rhs: &$t) -> $t {
                init();
                let (a, b, mut o) = (self.to_raw(), rhs.to_raw(), $t::zero_raw());
                check(unsafe { ffi::$add(&a, &b, &mut o, 1) });
                $t::from_raw(&o)
            }
        }
        impl Add<$t> for $t { type Output = $t; fn add(self, rhs: $t) -> $t { &self + &rhs } }
        impl<'a, F: FieldSpec> Mul<&'a PrimeFieldElem<F>> for &'a $t { // impl_scalar_mul_point! (macros.rs:1-32)
            type Output = $t;
            fn mul(self, k: &PrimeFieldElem<F>) -> $t {
                init();
                let (p, mut o) = (self.to_raw(), $t::zero_raw());
                check(unsafe { ffi::$mul(&p, k.limbs.as_ptr(), F::LIMBS as i32, &mut o, 1) });
                $t::from_raw(&o)
            }
        }
    };
}
point_ops!(G1Point, zkt_g1_affine, zkt_g1_add_batch, zkt_g1_neg_batch, zkt_g1_mul_batch);
point_ops!(G2Point, zkt_g2_affine, zkt_g2_add_batch, zkt_g2_neg_batch, zkt_g2_mul_batch);
point_ops!(SecpPoint, zkt_secp_affine, zkt_secp_add_batch, zkt_secp_add_batch, zkt_secp_mul_batch);

impl<'a> Neg for &'a G1Point { // g1_point.rs:177-195
    type Output = G1Point;
    fn neg(self) -> G1Point { init(); let (p, mut o) = (self.to_raw(), G1Point::zero_raw()); check(unsafe { ffi::zkt_g1_neg_batch(&p, &mut o, 1) }); G1Point::from_raw(&o) }
}
impl<'a> Neg for &'a G2Point {
    type Output = G2Point;
    fn neg(self) -> G2Point { init(); let (p, mut o) = (self.to_raw(), G2Point::zero_raw()); check(unsafe { ffi::zkt_g2_neg_batch(&p, &mut o, 1) }); G2Point::from_raw(&o) }
}

/// `AffinePoint` / `AffinePoints` of the reference's secp256k1 module (curves/secp256k1/{affine_point.rs, affine_points.rs:10-144}): the names the
/// Bulletproofs code is written against.  `sum` and the two products are one batched library call each.
pub type AffinePoint = SecpPoint;
#[derive(Clone, Debug)]
pub struct AffinePoints { pub points: Vec<AffinePoint> }
impl AffinePoints {
    pub fn new(points: &Vec<AffinePoint>) -> Self { AffinePoints { points: points.clone() } } // affine_points.rs:19-23
    pub fn len(&self) -> usize { self.points.len() }
    pub fn is_empty(&self) -> bool { self.points.is_empty() }
    /// :25-31 — the fold from AffinePoint::zero(): an empty vector sums to the point at infinity
    pub fn sum(&self) -> AffinePoint {
        init();
        let raw: Vec<zkt_secp_affine> = self.points.iter().map(|p| p.to_raw()).collect();
        let mut o = SecpPoint::zero_raw();
        check(unsafe { ffi::zkt_secp_sum(raw.as_ptr(), raw.len(), &mut o) });
        SecpPoint::from_raw(&o)
    }
    pub fn from(&self, idx: usize) -> Self { AffinePoints { points: self.points[idx..].to_vec() } } // :50-60
    pub fn to(&self, idx: usize) -> Self { AffinePoints { points: self.points[..idx].to_vec() } } // :62-72
    /// :105-122 — every point times ONE scalar
    pub fn scale<F: FieldSpec>(&self, k: &PrimeFieldElem<F>) -> Self {
        init();
        let raw: Vec<zkt_secp_affine> = self.points.iter().map(|p| p.to_raw()).collect();
        let mut out = vec![SecpPoint::zero_raw(); raw.len()];
        check(unsafe { ffi::zkt_secp_scale_batch(raw.as_ptr(), k.limbs.as_ptr(), F::LIMBS as i32, out.as_mut_ptr(), raw.len()) });
        AffinePoints { points: out.iter().map(SecpPoint::from_raw).collect() }
    }
    /// :124-144 — point i times scalar i
    pub fn mul_each<F: FieldSpec>(&self, ks: &[PrimeFieldElem<F>]) -> Self {
        init();
        if self.points.len() != ks.len() { panic!("Tried to multiply PrimeFieldElems of different size to AffinePoints"); }
        let raw: Vec<zkt_secp_affine> = self.points.iter().map(|p| p.to_raw()).collect();
        let k = PrimeFieldElem::<F>::flatten(ks);
        let mut out = vec![SecpPoint::zero_raw(); raw.len()];
        check(unsafe { ffi::zkt_secp_mul_batch(raw.as_ptr(), k.as_ptr(), F::LIMBS as i32, out.as_mut_ptr(), raw.len()) });
        AffinePoints { points: out.iter().map(SecpPoint::from_raw).collect() }
    }
}
impl<'a> Add<&'a AffinePoints> for &'a AffinePoints { // :84-103
    type Output = AffinePoints;
    fn add(self, rhs: &AffinePoints) -> AffinePoints {
        init();
        if self.len() != rhs.len() { panic!("Tried to add AffinePoints of diffrent length"); }
        let (a, b): (Vec<zkt_secp_affine>, Vec<zkt_secp_affine>) = (self.points.iter().map(|p| p.to_raw()).collect(), rhs.points.iter().map(|p| p.to_raw()).collect());
        let mut out = vec![SecpPoint::zero_raw(); a.len()];
        check(unsafe { ffi::zkt_secp_add_batch(a.as_ptr(), b.as_ptr(), out.as_mut_ptr(), a.len()) });
        AffinePoints { points: out.iter().map(SecpPoint::from_raw).collect() }
    }
}
impl<'a, F: FieldSpec> Mul<&'a PrimeFieldElem<F>> for &'a AffinePoints { type Output = AffinePoints; fn mul(self, k: &PrimeFieldElem<F>) -> AffinePoints { self.scale(k) } }
impl<'a, F: FieldSpec> Mul<&'a crate::field::PrimeFieldElems<F>> for &'a AffinePoints { type Output = AffinePoints; fn mul(self, ks: &crate::field::PrimeFieldElems<F>) -> AffinePoints { self.mul_each(&ks.0) } }
