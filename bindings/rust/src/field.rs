//! PrimeFieldElem over the four prime fields the reference instantiates (prime_field_elem.rs:263-457).
//! The reference keeps `{ f: Arc<PrimeField>, e: BigUint }`; here the field is a type parameter and `e` is a canonical limb vector.
use crate::{check, ffi, init};
use num_bigint::{BigUint, ToBigUint};
use num_traits::Zero as NumZero;
use std::marker::PhantomData;
use std::ops::{Add, Mul, Neg, Sub};

type Bin = unsafe extern "C" fn(*const u64, *const u64, *mut u64, usize) -> i32;
type Un = unsafe extern "C" fn(*const u64, *mut u64, usize) -> i32;
type Pow = unsafe extern "C" fn(*const u64, *const u64, usize, i32, *mut u64, usize) -> i32;
type Seq = unsafe extern "C" fn(*const u64, usize, *mut u64) -> i32;

/// one of the reference's `PrimeField`s: its order lives in the library, its entry points here
pub trait FieldSpec: Clone + std::fmt::Debug + PartialEq + Eq {
    const LIMBS: usize;
    const ADD: Bin; const SUB: Bin; const MUL: Bin;
    const SQR: Un; const CUBE: Un; const NEG: Un; const INV: Un;
    const POW: Pow; const POW_SEQ: Seq; const REPEAT: Seq;
    fn order() -> BigUint;
}
macro_rules! field_spec {
    ($name:ident, $limbs:expr, $p:ident, $order:expr) => {
        #[derive(Clone, Debug, PartialEq, Eq)]
        pub struct $name;
        paste_field!($name, $limbs, $p, $order);
    };
}
// no proc-macro dependencies: the entry-point names are spelled out
macro_rules! paste_field {
    ($name:ident, $limbs:expr, fq, $order:expr) => { impl_spec!($name, $limbs, $order, zkt_fq_add_batch, zkt_fq_sub_batch, zkt_fq_mul_batch, zkt_fq_sqr_batch, zkt_fq_cube_batch, zkt_fq_neg_batch, zkt_fq_inv_batch, zkt_fq_pow_batch, zkt_fq_pow_seq, zkt_fq_repeat); };
    ($name:ident, $limbs:expr, fr, $order:expr) => { impl_spec!($name, $limbs, $order, zkt_fr_add_batch, zkt_fr_sub_batch, zkt_fr_mul_batch, zkt_fr_sqr_batch, zkt_fr_cube_batch, zkt_fr_neg_batch, zkt_fr_inv_batch, zkt_fr_pow_batch, zkt_fr_pow_seq, zkt_fr_repeat); };
    ($name:ident, $limbs:expr, sp, $order:expr) => { impl_spec!($name, $limbs, $order, zkt_sp_add_batch, zkt_sp_sub_batch, zkt_sp_mul_batch, zkt_sp_sqr_batch, zkt_sp_cube_batch, zkt_sp_neg_batch, zkt_sp_inv_batch, zkt_sp_pow_batch, zkt_sp_pow_seq, zkt_sp_repeat); };
    ($name:ident, $limbs:expr, sn, $order:expr) => { impl_spec!($name, $limbs, $order, zkt_sn_add_batch, zkt_sn_sub_batch, zkt_sn_mul_batch, zkt_sn_sqr_batch, zkt_sn_cube_batch, zkt_sn_neg_batch, zkt_sn_inv_batch, zkt_sn_pow_batch, zkt_sn_pow_seq, zkt_sn_repeat); };
}
macro_rules! impl_spec {
    ($name:ident, $limbs:expr, $order:expr, $add:ident, $sub:ident, $mul:ident, $sqr:ident, $cube:ident, $neg:ident, $inv:ident, $pow:ident, $seq:ident, $rep:ident) => {
        impl FieldSpec for $name {
            const LIMBS: usize = $limbs;
            const ADD: Bin = ffi::$add; const SUB: Bin = ffi::$sub; const MUL: Bin = ffi::$mul;
            const SQR: Un = ffi::$sqr; const CUBE: Un = ffi::$cube; const NEG: Un = ffi::$neg; const INV: Un = ffi::$inv;
            const POW: Pow = ffi::$pow; const POW_SEQ: Seq = ffi::$seq; const REPEAT: Seq = ffi::$rep;
            fn order() -> BigUint { BigUint::parse_bytes($order, 16).unwrap() }
        }
    };
}
field_spec!(Bls12Q, 6, fq, b"1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab"); // params.rs:9
field_spec!(Bls12R, 4, fr, b"73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001"); // params.rs:14
field_spec!(SecpP, 4, sp, b"fffffffffffffffffffffffffffffffffffffffffffffffffffffffefffffc2f"); // secp256k1/affine_point.rs:30-33
field_spec!(SecpN, 4, sn, b"fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364141");

/// `PrimeFieldElem` (prime_field_elem.rs:57-61): `e` as little-endian u64 limbs, always the canonical residue
#[derive(Clone, Debug, PartialEq, Eq)]
pub struct PrimeFieldElem<F: FieldSpec> {
    pub limbs: Vec<u64>,
    _f: PhantomData<F>,
}
pub type Fq1 = PrimeFieldElem<Bls12Q>; // fq1.rs:13
pub type Fr = PrimeFieldElem<Bls12R>;
pub type SecpFq = PrimeFieldElem<SecpP>;
pub type SecpFr = PrimeFieldElem<SecpN>;

pub(crate) fn to_limbs(v: &BigUint, n: usize) -> Vec<u64> {
    let mut l = v.to_u64_digits();
    l.resize(n, 0);
    l
}
pub(crate) fn from_limbs(l: &[u64]) -> BigUint {
    let mut bytes = Vec::with_capacity(l.len() * 8);
    for w in l { bytes.extend_from_slice(&w.to_le_bytes()); }
    BigUint::from_bytes_le(&bytes)
}

impl<F: FieldSpec> PrimeFieldElem<F> {
    /// `PrimeFieldElem::new` (prime_field_elem.rs:263-272): reduces `e` mod the order.  Values wider than the field are brought below
    /// 2^(64*LIMBS) on the host (a marshalling step, `%` on BigUint as in the reference); the library reduces any limb vector on load.
    pub fn new(e: &impl ToBigUint) -> Self {
        init();
        let mut v = e.to_biguint().expect("unsigned value");
        if v.bits() as usize > 64 * F::LIMBS { v %= F::order(); }
        let raw = to_limbs(&v, F::LIMBS);
        // canonicalise through the library: x + 0
        let zero = vec![0u64; F::LIMBS];
        let mut out = vec![0u64; F::LIMBS];
        check(unsafe { F::ADD(raw.as_ptr(), zero.as_ptr(), out.as_mut_ptr(), 1) });
        PrimeFieldElem { limbs: out, _f: PhantomData }
    }
    pub(crate) fn from_canonical_limbs(l: &[u64]) -> Self { PrimeFieldElem { limbs: l.to_vec(), _f: PhantomData } }
    pub fn e(&self) -> BigUint { from_limbs(&self.limbs) }
    pub fn is_zero(&self) -> bool { self.limbs.iter().all(|w| *w == 0) }

    fn bin(op: Bin, a: &Self, b: &Self) -> Self {
        let mut out = vec![0u64; F::LIMBS];
        check(unsafe { op(a.limbs.as_ptr(), b.limbs.as_ptr(), out.as_mut_ptr(), 1) });
        Self::from_canonical_limbs(&out)
    }
    fn un(op: Un, a: &Self) -> Self {
        let mut out = vec![0u64; F::LIMBS];
        check(unsafe { op(a.limbs.as_ptr(), out.as_mut_ptr(), 1) });
        Self::from_canonical_limbs(&out)
    }
    pub fn plus(&self, rhs: &impl ToBigUint) -> Self { Self::bin(F::ADD, self, &Self::new(rhs)) } // :278-286
    pub fn minus(&self, rhs: &impl ToBigUint) -> Self { Self::bin(F::SUB, self, &Self::new(rhs)) } // :288-300
    pub fn times(&self, rhs: &impl ToBigUint) -> Self { Self::bin(F::MUL, self, &Self::new(rhs)) } // :302-308
    pub fn sq(&self) -> Self { Self::un(F::SQR, self) } // :330-335
    pub fn cube(&self) -> Self { Self::un(F::CUBE, self) } // :337-344
    pub fn negate(&self) -> Self { Self::un(F::NEG, self) } // :448-457
    /// :311-328
    pub fn pow(&self, rhs: &impl ToBigUint) -> Self {
        let e = rhs.to_biguint().expect("unsigned exponent").to_u64_digits();
        let e = if e.is_empty() { vec![0u64] } else { e };
        let mut out = vec![0u64; F::LIMBS];
        check(unsafe { F::POW(self.limbs.as_ptr(), e.as_ptr(), e.len(), 1, out.as_mut_ptr(), 1) });
        Self::from_canonical_limbs(&out)
    }
    /// :346-361 — 1, x, x^2, ..., x^(n-1)
    pub fn pow_seq(&self, n: usize) -> Vec<Self> { Self::seq(F::POW_SEQ, self, n) }
    /// :363-376
    pub fn repeat(&self, n: usize) -> Vec<Self> { Self::seq(F::REPEAT, self, n) }
    fn seq(op: Seq, a: &Self, n: usize) -> Vec<Self> {
        let mut out = vec![0u64; F::LIMBS * n];
        check(unsafe { op(a.limbs.as_ptr(), n, out.as_mut_ptr()) });
        out.chunks(F::LIMBS).map(Self::from_canonical_limbs).collect()
    }
    /// :379-432 — Err on zero
    pub fn safe_inv(&self) -> Result<Self, String> {
        if self.is_zero() { return Err("Cannot find inverse of zero".to_string()); }
        Ok(Self::un(F::INV, self))
    }
    /// :434-436 — panics on zero
    pub fn inv(&self) -> Self { self.safe_inv().unwrap() }

    // ---- the batch forms a hot loop should use: n elements, one launch ----
    pub fn flatten(xs: &[Self]) -> Vec<u64> { xs.iter().flat_map(|x| x.limbs.iter().copied()).collect() }
    pub fn unflatten(l: &[u64]) -> Vec<Self> { l.chunks(F::LIMBS).map(Self::from_canonical_limbs).collect() }
    pub fn mul_batch(a: &[Self], b: &[Self]) -> Vec<Self> { Self::bin_batch(F::MUL, a, b) }
    pub fn add_batch(a: &[Self], b: &[Self]) -> Vec<Self> { Self::bin_batch(F::ADD, a, b) }
    pub fn sub_batch(a: &[Self], b: &[Self]) -> Vec<Self> { Self::bin_batch(F::SUB, a, b) }
    pub fn inv_batch(a: &[Self]) -> Vec<Self> {
        let fa = Self::flatten(a);
        let mut out = vec![0u64; fa.len()];
        check(unsafe { F::INV(fa.as_ptr(), out.as_mut_ptr(), a.len()) });
        Self::unflatten(&out)
    }
    fn bin_batch(op: Bin, a: &[Self], b: &[Self]) -> Vec<Self> {
        assert_eq!(a.len(), b.len(), "Tried to operate on vectors of different length"); // prime_field_elems.rs:96
        let (fa, fb) = (Self::flatten(a), Self::flatten(b));
        let mut out = vec![0u64; fa.len()];
        check(unsafe { op(fa.as_ptr(), fb.as_ptr(), out.as_mut_ptr(), a.len()) });
        Self::unflatten(&out)
    }
}

impl<F: FieldSpec> ToBigUint for PrimeFieldElem<F> {
    fn to_biguint(&self) -> Option<BigUint> { Some(self.e()) }
}
impl<F: FieldSpec> NumZero for PrimeFieldElem<F> {
    fn zero() -> Self { init(); Self::from_canonical_limbs(&vec![0u64; F::LIMBS]) }
    fn is_zero(&self) -> bool { PrimeFieldElem::is_zero(self) }
}
macro_rules! impl_op {
    ($tr:ident, $f:ident, $op:ident) => {
        impl<'a, F: FieldSpec> $tr<&'a PrimeFieldElem<F>> for &'a PrimeFieldElem<F> {
            type Output = PrimeFieldElem<F>;
            fn $f(self, rhs: &PrimeFieldElem<F>) -> PrimeFieldElem<F> { PrimeFieldElem::bin(F::$op, self, rhs) }
        }
        impl<F: FieldSpec> $tr<PrimeFieldElem<F>> for PrimeFieldElem<F> {
            type Output = PrimeFieldElem<F>;
            fn $f(self, rhs: PrimeFieldElem<F>) -> PrimeFieldElem<F> { PrimeFieldElem::bin(F::$op, &self, &rhs) }
        }
    };
}
impl_op!(Add, add, ADD);
impl_op!(Sub, sub, SUB);
impl_op!(Mul, mul, MUL);
impl<F: FieldSpec> Neg for PrimeFieldElem<F> {
    type Output = PrimeFieldElem<F>;
    fn neg(self) -> PrimeFieldElem<F> { self.negate() }
}
impl<'a, F: FieldSpec> Neg for &'a PrimeFieldElem<F> {
    type Output = PrimeFieldElem<F>;
    fn neg(self) -> PrimeFieldElem<F> { self.negate() }
}
