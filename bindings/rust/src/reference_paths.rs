//! The reference crate's OWN module paths and its runtime-order field types, over the typed shim.
//!
//! The reference passes a field around as a VALUE — `PrimeField { order: BigUint }` (building_block/field/prime_field.rs:15-33), elements carry
//! `{ f: Arc<PrimeField>, e: BigUint }` (prime_field_elem.rs:57-61) — and its call sites import types by module path
//! (`use crate::building_block::curves::bls12_381::g1_point::G1Point;`).  The typed modules of this crate (`field`, `points`, `groth16`, ...) bind a field
//! at compile time (`PrimeFieldElem<Bls12R>`), which is what the FFI wants but not what reference code spells.  This module gives a maintainer both halves:
//!   * `building_block::...` and `zk::...`: the reference's paths, re-exporting the shim's types, so `use` lines of the reference keep compiling;
//!   * `building_block::field::{prime_field::PrimeField, prime_field_elem::PrimeFieldElem, prime_field_elems::PrimeFieldElems, sparse_vec::SparseVec}`:
//!     NON-generic types with the reference's fields and constructor signatures (`PrimeFieldElem::new(f: &Arc<PrimeField>, e: &impl ToBigUint)`), which look the
//!     order up at run time among the four fields the engine instantiates (BLS12-381 q and r, secp256k1 p and n) and forward to the typed element.
//!     Any other order (the reference's unit tests use 97, 53, 11) panics: those fields stay on the reference's CPU type.
//! Conversions to the typed forms (`typed::<Bls12R>()`, `From`) are what the protocol structs take; INTEGRATION.md lists the edits a maintainer makes.
use crate::field as typed;
use num_bigint::{BigUint, ToBigUint};
use std::sync::Arc;

/// which of the engine's four fields an order names
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum FieldId { Bls12Q, Bls12R, SecpP, SecpN }

fn field_id(order: &BigUint) -> FieldId {
    use typed::FieldSpec;
    if *order == typed::Bls12Q::order() { FieldId::Bls12Q }
    else if *order == typed::Bls12R::order() { FieldId::Bls12R }
    else if *order == typed::SecpP::order() { FieldId::SecpP }
    else if *order == typed::SecpN::order() { FieldId::SecpN }
    else { panic!("field order {} is not one of the four fields the MI355X engine instantiates (BLS12-381 q / r, secp256k1 p / n)", order) }
}

/// run `$body` with `$T` bound to the typed field spec of `$id`
macro_rules! with_field {
    ($id:expr, $T:ident, $body:expr) => {
        match $id {
            FieldId::Bls12Q => { type $T = typed::Bls12Q; $body }
            FieldId::Bls12R => { type $T = typed::Bls12R; $body }
            FieldId::SecpP => { type $T = typed::SecpP; $body }
            FieldId::SecpN => { type $T = typed::SecpN; $body }
        }
    };
}

pub mod building_block {
    pub mod field {
        pub mod prime_field {
            use super::super::super::*;
            use super::prime_field_elem::PrimeFieldElem;
            use super::prime_field_elems::PrimeFieldElems;
            /// prime_field.rs:15-18
            #[derive(Debug, Clone, Hash, PartialEq, Eq)]
            pub struct PrimeField { order: BigUint }
            impl PrimeField {
                pub fn new(order: &impl ToBigUint) -> Self { PrimeField { order: order.to_biguint().expect("unsigned order") } } // :21-25
                pub fn order(&self) -> BigUint { self.order.clone() } // :27-29
                pub fn order_ref(&self) -> &BigUint { &self.order } // :31-33
                pub fn id(&self) -> FieldId { field_id(&self.order) }
                pub fn elem(&self, x: &impl ToBigUint) -> PrimeFieldElem { PrimeFieldElem::new(&Arc::new(self.clone()), x) } // :35-38
                pub fn repeated_elem(&self, x: &impl ToBigUint, count: usize) -> PrimeFieldElems { // :56-60
                    PrimeFieldElems((0..count).map(|_| self.elem(x)).collect())
                }
                pub fn first_n_powers_of_x(&self, x: &impl ToBigUint, n: usize) -> PrimeFieldElems { PrimeFieldElems(self.elem(x).pow_seq(n)) } // :62-70: 1, x, .., x^(n-1)
                /// :73-85 — ceil(bits / 8) bytes of OS entropy, big-endian, reduced; redrawn while zero if `exclude_zero`
                pub fn rand_elem(&self, exclude_zero: bool) -> PrimeFieldElem {
                    let f = Arc::new(self.clone());
                    with_field!(self.id(), T, PrimeFieldElem::from_typed(&f, &typed::PrimeField::<T>::new().rand_elem(exclude_zero)))
                }
                pub fn rand_elems(&self, n: &usize, exclude_zero: bool) -> PrimeFieldElems { PrimeFieldElems((0..*n).map(|_| self.rand_elem(exclude_zero)).collect()) } // :87-90
                /// the typed field value the protocol structs of this crate take (`groth16::CRS::new(&f.typed::<Bls12R>(), ..)`); panics if the orders differ
                pub fn typed<T: typed::FieldSpec>(&self) -> typed::PrimeField<T> { assert_eq!(T::order(), self.order, "field order mismatch"); typed::PrimeField::<T>::new() }
            }
        }
        pub mod prime_field_elem {
            use super::super::super::*;
            use super::prime_field::PrimeField;
            use std::ops::{Add, Mul, Neg, Sub};
            /// prime_field_elem.rs:57-61: the reference's two fields, `e` always the canonical residue
            #[derive(Debug, Clone, PartialEq, Eq)]
            pub struct PrimeFieldElem { pub f: Arc<PrimeField>, pub e: BigUint }
            impl PrimeFieldElem {
                /// :263-272 — reduces `e` mod the order
                pub fn new(f: &Arc<PrimeField>, e: &impl ToBigUint) -> Self { with_field!(f.id(), T, Self::from_typed(f, &typed::PrimeFieldElem::<T>::new(e))) }
                pub fn from_typed<T: typed::FieldSpec>(f: &Arc<PrimeField>, x: &typed::PrimeFieldElem<T>) -> Self { PrimeFieldElem { f: f.clone(), e: x.e() } }
                /// the typed element the batched entry points take; panics if the orders differ
                pub fn typed<T: typed::FieldSpec>(&self) -> typed::PrimeFieldElem<T> { assert_eq!(T::order(), *self.f.order_ref(), "field order mismatch"); typed::PrimeFieldElem::<T>::new(&self.e) }
                pub fn is_zero(&self) -> bool { self.e == BigUint::from(0u8) }
                fn un(&self, op: impl Fn(FieldId, &BigUint) -> BigUint) -> Self { PrimeFieldElem { f: self.f.clone(), e: op(self.f.id(), &self.e) } }
                pub fn plus(&self, rhs: &impl ToBigUint) -> Self { let r = rhs.to_biguint().unwrap(); self.un(|id, e| with_field!(id, T, typed::PrimeFieldElem::<T>::new(e).plus(&r).e())) } // :278-286
                pub fn minus(&self, rhs: &impl ToBigUint) -> Self { let r = rhs.to_biguint().unwrap(); self.un(|id, e| with_field!(id, T, typed::PrimeFieldElem::<T>::new(e).minus(&r).e())) } // :288-300
                pub fn times(&self, rhs: &impl ToBigUint) -> Self { let r = rhs.to_biguint().unwrap(); self.un(|id, e| with_field!(id, T, typed::PrimeFieldElem::<T>::new(e).times(&r).e())) } // :302-308
                pub fn pow(&self, rhs: &impl ToBigUint) -> Self { let r = rhs.to_biguint().unwrap(); self.un(|id, e| with_field!(id, T, typed::PrimeFieldElem::<T>::new(e).pow(&r).e())) } // :311-328
                pub fn sq(&self) -> Self { self.un(|id, e| with_field!(id, T, typed::PrimeFieldElem::<T>::new(e).sq().e())) } // :330-335
                pub fn cube(&self) -> Self { self.un(|id, e| with_field!(id, T, typed::PrimeFieldElem::<T>::new(e).cube().e())) } // :337-344
                pub fn negate(&self) -> Self { self.un(|id, e| with_field!(id, T, typed::PrimeFieldElem::<T>::new(e).negate().e())) } // :448-457
                pub fn pow_seq(&self, n: usize) -> Vec<Self> { with_field!(self.f.id(), T, typed::PrimeFieldElem::<T>::new(&self.e).pow_seq(n).iter().map(|x| Self::from_typed(&self.f, x)).collect()) } // :346-361
                pub fn repeat(&self, n: usize) -> Vec<Self> { (0..n).map(|_| self.clone()).collect() } // :363-376
                pub fn safe_inv(&self) -> Result<Self, String> { // :379-432
                    if self.is_zero() { return Err("Cannot find inverse of zero".to_string()); }
                    Ok(self.un(|id, e| with_field!(id, T, typed::PrimeFieldElem::<T>::new(e).inv().e())))
                }
                pub fn inv(&self) -> Self { self.safe_inv().unwrap() } // :434-436
            }
            impl ToBigUint for PrimeFieldElem { fn to_biguint(&self) -> Option<BigUint> { Some(self.e.clone()) } } // to_biguint.rs
            macro_rules! op { ($tr:ident, $f:ident, $m:ident) => {
                impl<'a> $tr<&'a PrimeFieldElem> for &'a PrimeFieldElem { type Output = PrimeFieldElem; fn $f(self, rhs: &PrimeFieldElem) -> PrimeFieldElem { self.$m(rhs) } } // :96-188
                impl $tr<PrimeFieldElem> for PrimeFieldElem { type Output = PrimeFieldElem; fn $f(self, rhs: PrimeFieldElem) -> PrimeFieldElem { self.$m(&rhs) } }
            } }
            op!(Add, add, plus); op!(Sub, sub, minus); op!(Mul, mul, times);
            impl Neg for PrimeFieldElem { type Output = PrimeFieldElem; fn neg(self) -> PrimeFieldElem { self.negate() } }
            impl<'a> Neg for &'a PrimeFieldElem { type Output = PrimeFieldElem; fn neg(self) -> PrimeFieldElem { self.negate() } }
        }
        pub mod prime_field_elems {
            use super::super::super::*;
            use super::prime_field_elem::PrimeFieldElem;
            /// prime_field_elems.rs:13-175 — element-wise `+ - *`, `* scalar`, `sum`: each ONE batched call through the typed vector
            #[derive(Debug, Clone, PartialEq, Eq)]
            pub struct PrimeFieldElems(pub Vec<PrimeFieldElem>);
            impl PrimeFieldElems {
                pub fn new(xs: &[PrimeFieldElem]) -> Self { PrimeFieldElems(xs.to_vec()) } // :30-32
                pub fn len(&self) -> usize { self.0.len() }
                pub fn typed<T: typed::FieldSpec>(&self) -> typed::PrimeFieldElems<T> { typed::PrimeFieldElems(self.0.iter().map(|x| x.typed::<T>()).collect()) }
                pub fn from_typed<T: typed::FieldSpec>(f: &Arc<super::prime_field::PrimeField>, v: &typed::PrimeFieldElems<T>) -> Self { PrimeFieldElems(v.0.iter().map(|x| PrimeFieldElem::from_typed(f, x)).collect()) }
                pub fn sum(&self) -> PrimeFieldElem { // :35-41
                    assert!(self.0.len() > 0);
                    let f = self.0[0].f.clone();
                    with_field!(f.id(), T, PrimeFieldElem::from_typed(&f, &self.typed::<T>().sum()))
                }
            }
        }
        pub mod sparse_vec {
            use super::super::super::*;
            use super::prime_field::PrimeField;
            use super::prime_field_elem::PrimeFieldElem;
            /// sparse_vec.rs:15-20
            #[derive(Clone, Debug)]
            pub struct SparseVec { pub f: PrimeField, pub size: PrimeFieldElem, pub elems: std::collections::HashMap<PrimeFieldElem, PrimeFieldElem> }
            impl SparseVec {
                /// the statement-wire form the typed verifiers take (`groth16::Verifier::verify(.., &stmt.typed::<Bls12R>())`)
                pub fn typed<T: typed::FieldSpec>(&self) -> typed::SparseVec<T> {
                    let size = self.size.e.to_u64_digits().first().copied().unwrap_or(0) as usize;
                    let mut v = typed::SparseVec::<T>::new(size);
                    for (k, x) in &self.elems { v.set(k.e.to_u64_digits().first().copied().unwrap_or(0) as usize, &x.typed::<T>()); }
                    v
                }
            }
            impl std::hash::Hash for PrimeFieldElem { fn hash<H: std::hash::Hasher>(&self, h: &mut H) { self.e.hash(h) } } // prime_field_elem.rs:63-67: the key of the reference's map
        }
        pub mod polynomial { pub use crate::polynomial::{G1Bases, G2Bases, Polynomial}; }
    }
    pub mod curves {
        pub mod bls12_381 {
            pub mod fq1 { pub use crate::field::Fq1; }
            pub mod fq2 { pub use crate::tower::Fq2; }
            pub mod fq6 { pub use crate::tower::Fq6; }
            pub mod fq12 { pub use crate::tower::Fq12; }
            pub mod g1_point { pub use crate::points::G1Point; }
            pub mod g2_point { pub use crate::points::G2Point; }
            pub mod gt_point { pub use crate::pairing::GTPoint; }
            pub mod pairing { pub use crate::pairing::Pairing; }
            pub mod private_key { pub use crate::signature::PrivateKey; }
            pub mod signature { pub use crate::signature::Signer; }
        }
        pub mod secp256k1 {
            pub mod affine_point { pub use crate::points::AffinePoint; }
            pub mod affine_points { pub use crate::points::AffinePoints; }
        }
    }
}
pub mod zk {
    pub mod w_trusted_setup {
        pub mod groth16 {
            pub mod zktoolkit_based {
                pub mod crs { pub use crate::groth16::{CRS, G1, G2, GT}; }
                pub mod proof { pub use crate::groth16::Proof; }
                pub mod prover { pub use crate::groth16::Prover; }
                pub mod verifier { pub use crate::groth16::Verifier; }
            }
        }
        pub mod pinocchio {
            pub mod crs { pub use crate::pinocchio::CRS; }
            pub mod proof { pub use crate::pinocchio::Proof; }
            pub mod prover { pub use crate::pinocchio::Prover; }
            pub mod verifier { pub use crate::pinocchio::Verifier; }
        }
    }
    pub mod wo_trusted_setup {
        pub mod bulletproofs { pub use crate::bulletproofs::Bulletproofs; }
    }
}
