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
type Sum = unsafe extern "C" fn(*const u64, usize, *mut u64) -> i32;

/// one of the reference's `PrimeField`s: its order lives in the library, its entry points here
pub trait FieldSpec: Clone + std::fmt::Debug + PartialEq + Eq {
    const LIMBS: usize;
    const ADD: Bin; const SUB: Bin; const MUL: Bin;
    const SQR: Un; const CUBE: Un; const NEG: Un; const INV: Un;
    const POW: Pow; const POW_SEQ: Seq; const REPEAT: Seq;
    const SUM: Sum; const SCALE: Bin;
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
    ($name:ident, $limbs:expr, fq, $order:expr) => { impl_spec!($name, $limbs, $order, zkt_fq_add_batch, zkt_fq_sub_batch, zkt_fq_mul_batch, zkt_fq_sqr_batch, zkt_fq_cube_batch, zkt_fq_neg_batch, zkt_fq_inv_batch, zkt_fq_pow_batch, zkt_fq_pow_seq, zkt_fq_repeat, zkt_fq_sum, zkt_fq_scale_batch); };
    ($name:ident, $limbs:expr, fr, $order:expr) => { impl_spec!($name, $limbs, $order, zkt_fr_add_batch, zkt_fr_sub_batch, zkt_fr_mul_batch, zkt_fr_sqr_batch, zkt_fr_cube_batch, zkt_fr_neg_batch, zkt_fr_inv_batch, zkt_fr_pow_batch, zkt_fr_pow_seq, zkt_fr_repeat, zkt_fr_sum, zkt_fr_scale_batch); };
    ($name:ident, $limbs:expr, sp, $order:expr) => { impl_spec!($name, $limbs, $order, zkt_sp_add_batch, zkt_sp_sub_batch, zkt_sp_mul_batch, zkt_sp_sqr_batch, zkt_sp_cube_batch, zkt_sp_neg_batch, zkt_sp_inv_batch, zkt_sp_pow_batch, zkt_sp_pow_seq, zkt_sp_repeat, zkt_sp_sum, zkt_sp_scale_batch); };
    ($name:ident, $limbs:expr, sn, $order:expr) => { impl_spec!($name, $limbs, $order, zkt_sn_add_batch, zkt_sn_sub_batch, zkt_sn_mul_batch, zkt_sn_sqr_batch, zkt_sn_cube_batch, zkt_sn_neg_batch, zkt_sn_inv_batch, zkt_sn_pow_batch, zkt_sn_pow_seq, zkt_sn_repeat, zkt_sn_sum, zkt_sn_scale_batch); };
}
macro_rules! impl_spec {
    ($name:ident, $limbs:expr, $order:expr, $add:ident, $sub:ident, $mul:ident, $sqr:ident, $cube:ident, $neg:ident, $inv:ident, $pow:ident, $seq:ident, $rep:ident, $sum:ident, $scale:ident) => {
        impl FieldSpec for $name {
            const LIMBS: usize = $limbs;
            const ADD: Bin = ffi::$add; const SUB: Bin = ffi::$sub; const MUL: Bin = ffi::$mul;
            const SQR: Un = ffi::$sqr; const CUBE: Un = ffi::$cube; const NEG: Un = ffi::$neg; const INV: Un = ffi::$inv;
            const POW: Pow = ffi::$pow; const POW_SEQ: Seq = ffi::$seq; const REPEAT: Seq = ffi::$rep;
            const SUM: Sum = ffi::$sum; const SCALE: Bin = ffi::$scale;
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

/// `PrimeField` (building_block/field/prime_field.rs:14-100): the field as a value, as the reference passes it around.  The order lives in the
/// library; what a caller needs from the value are `elem` and the entropy draws of `rand_elem` (prime_field.rs:73-85: ceil(bits / 8) random
/// big-endian bytes reduced mod the order, redrawn while zero if `exclude_zero`).  Entropy: the OS source (`/dev/urandom`), as `RandomNumber`
/// (building_block/random_number.rs:8-13) seeds from it.
#[derive(Clone, Debug, PartialEq, Eq)]
pub struct PrimeField<F: FieldSpec> { _f: PhantomData<F> }
impl<F: FieldSpec> PrimeField<F> {
    pub fn new() -> Self { PrimeField { _f: PhantomData } }
    pub fn order(&self) -> BigUint { F::order() }
    pub fn elem(&self, x: &impl ToBigUint) -> PrimeFieldElem<F> { PrimeFieldElem::new(x) } // prime_field.rs:44-46
    pub fn rand_elem(&self, exclude_zero: bool) -> PrimeFieldElem<F> {
        use std::io::Read;
        let buf_size = ((F::order().bits() as f64) / 8f64).ceil() as usize;
        let mut buf = vec![0u8; buf_size];
        loop {
            std::fs::File::open("/dev/urandom").and_then(|mut f| f.read_exact(&mut buf)).expect("OS entropy");
            let x = PrimeFieldElem::<F>::new(&BigUint::from_bytes_be(&buf));
            if !exclude_zero || !x.is_zero() { return x; }
        }
    }
    pub fn rand_elems(&self, n: &usize, exclude_zero: bool) -> PrimeFieldElems<F> { // prime_field.rs:87-90
        PrimeFieldElems((0..*n).map(|_| self.rand_elem(exclude_zero)).collect())
    }
}

/// `PrimeFieldElems` (building_block/field/prime_field_elems.rs:13-175): a vector of field elements with element-wise `+ - *`, `* scalar`
/// and `sum` — each ONE batched call into the library.
#[derive(Clone, Debug, PartialEq, Eq)]
pub struct PrimeFieldElems<F: FieldSpec>(pub Vec<PrimeFieldElem<F>>);
impl<F: FieldSpec> PrimeFieldElems<F> {
    pub fn new(xs: &[PrimeFieldElem<F>]) -> Self { PrimeFieldElems(xs.to_vec()) } // :30-32
    pub fn len(&self) -> usize { self.0.len() }
    pub fn is_empty(&self) -> bool { self.0.is_empty() }
    /// :35-41 — panics on an empty vector, as the reference's assert does
    pub fn sum(&self) -> PrimeFieldElem<F> {
        assert!(self.0.len() > 0);
        let fa = PrimeFieldElem::<F>::flatten(&self.0);
        let mut out = vec![0u64; F::LIMBS];
        check(unsafe { F::SUM(fa.as_ptr(), self.0.len(), out.as_mut_ptr()) });
        PrimeFieldElem::from_canonical_limbs(&out)
    }
    /// :43-54 / :56-67
    pub fn from(&self, idx: usize) -> Self { if idx >= self.len() { panic!("index outside the range is specified"); } PrimeFieldElems(self.0[idx..].to_vec()) }
    pub fn to(&self, idx: usize) -> Self { if idx > self.len() { panic!("index outside the range is specified"); } PrimeFieldElems(self.0[..idx].to_vec()) }
    /// :152-175 — every element times ONE scalar
    pub fn scale(&self, k: &PrimeFieldElem<F>) -> Self {
        assert!(self.len() > 0);
        let fa = PrimeFieldElem::<F>::flatten(&self.0);
        let mut out = vec![0u64; fa.len()];
        check(unsafe { F::SCALE(fa.as_ptr(), k.limbs.as_ptr(), out.as_mut_ptr(), self.len()) });
        PrimeFieldElems(PrimeFieldElem::unflatten(&out))
    }
}
impl<F: FieldSpec> std::ops::Index<usize> for PrimeFieldElems<F> { type Output = PrimeFieldElem<F>; fn index(&self, i: usize) -> &PrimeFieldElem<F> { &self.0[i] } }
macro_rules! impl_vec_op {
    ($tr:ident, $f:ident, $batch:ident) => {
        impl<'a, F: FieldSpec> $tr<&'a PrimeFieldElems<F>> for &'a PrimeFieldElems<F> { // prime_field_elems.rs:90-150
            type Output = PrimeFieldElems<F>;
            fn $f(self, rhs: &PrimeFieldElems<F>) -> PrimeFieldElems<F> { assert!(self.len() > 0 && self.len() == rhs.len()); PrimeFieldElems(PrimeFieldElem::$batch(&self.0, &rhs.0)) }
        }
    };
}
impl_vec_op!(Add, add, add_batch);
impl_vec_op!(Sub, sub, sub_batch);
impl_vec_op!(Mul, mul, mul_batch);
impl<'a, F: FieldSpec> Mul<&'a PrimeFieldElem<F>> for &'a PrimeFieldElems<F> { type Output = PrimeFieldElems<F>; fn mul(self, k: &PrimeFieldElem<F>) -> PrimeFieldElems<F> { self.scale(k) } }

/// `SparseVec` (building_block/field/sparse_vec.rs:15-31): size + the non-zero entries; what `Verifier::verify` takes as the statement wires
#[derive(Clone, Debug)]
pub struct SparseVec<F: FieldSpec> { pub size: usize, pub elems: std::collections::HashMap<usize, PrimeFieldElem<F>> }
impl<F: FieldSpec> SparseVec<F> {
    pub fn new(size: usize) -> Self { SparseVec { size, elems: std::collections::HashMap::new() } } // :53-64
    pub fn set(&mut self, index: usize, n: &PrimeFieldElem<F>) { if index >= self.size { panic!("Index {} is out of range. The size of vector is {}", index, self.size); } if !n.is_zero() { self.elems.insert(index, n.clone()); } } // :70-79
    pub fn get(&self, index: usize) -> PrimeFieldElem<F> { if index >= self.size { panic!("Index {} is out of range. The size of vector is {}", index, self.size); } self.elems.get(&index).cloned().unwrap_or_else(<PrimeFieldElem<F> as NumZero>::zero) } // :81-91
    pub fn to_dense(&self) -> Vec<PrimeFieldElem<F>> { (0..self.size).map(|i| self.get(i)).collect() }
}
