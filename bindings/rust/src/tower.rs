//! Fq2 = Fq[u]/(u^2+1), Fq6 = Fq2[v]/(v^3-(1+u)), Fq12 = Fq6[w]/(w^2-v): fq2.rs:15-151, fq6.rs:15-171, fq12.rs:17-172.
//! Struct fields keep the reference's order (high degree first); the limb image is exactly the C ABI's {u1,u0} / {v2,v1,v0} / {w1,w0}.
use crate::field::{Fq1, PrimeFieldElem};
use crate::{check, ffi, init};
use num_bigint::{BigUint, ToBigUint};
use std::ops::{Add, Mul, Neg, Sub};

#[derive(Clone, Debug, PartialEq, Eq)]
pub struct Fq2 { pub u1: Fq1, pub u0: Fq1 } // fq2.rs:16-19
#[derive(Clone, Debug, PartialEq, Eq)]
pub struct Fq6 { pub v2: Fq2, pub v1: Fq2, pub v0: Fq2 } // fq6.rs:16-20
#[derive(Clone, Debug, PartialEq, Eq)]
pub struct Fq12 { pub w1: Fq6, pub w0: Fq6 } // fq12.rs:18-21

pub trait Limbs: Sized {
    const N: usize; // u64 limbs
    fn write(&self, out: &mut Vec<u64>);
    fn read(l: &[u64]) -> Self;
    fn to_vec(&self) -> Vec<u64> { let mut v = Vec::with_capacity(Self::N); self.write(&mut v); v }
}
impl Limbs for Fq1 {
    const N: usize = 6;
    fn write(&self, out: &mut Vec<u64>) { out.extend_from_slice(&self.limbs); }
    fn read(l: &[u64]) -> Self { PrimeFieldElem::from_canonical_limbs(&l[..6]) }
}
impl Limbs for Fq2 {
    const N: usize = 12;
    fn write(&self, out: &mut Vec<u64>) { self.u1.write(out); self.u0.write(out); }
    fn read(l: &[u64]) -> Self { Fq2 { u1: Fq1::read(&l[0..6]), u0: Fq1::read(&l[6..12]) } }
}
impl Limbs for Fq6 {
    const N: usize = 36;
    fn write(&self, out: &mut Vec<u64>) { self.v2.write(out); self.v1.write(out); self.v0.write(out); }
    fn read(l: &[u64]) -> Self { Fq6 { v2: Fq2::read(&l[0..12]), v1: Fq2::read(&l[12..24]), v0: Fq2::read(&l[24..36]) } }
}
impl Limbs for Fq12 {
    const N: usize = 72;
    fn write(&self, out: &mut Vec<u64>) { self.w1.write(out); self.w0.write(out); }
    fn read(l: &[u64]) -> Self { Fq12 { w1: Fq6::read(&l[0..36]), w0: Fq6::read(&l[36..72]) } }
}

type Bin = unsafe extern "C" fn(*const u64, *const u64, *mut u64, usize) -> i32;
type Un = unsafe extern "C" fn(*const u64, *mut u64, usize) -> i32;
fn bin<T: Limbs>(op: Bin, a: &T, b: &T) -> T {
    init();
    let (fa, fb) = (a.to_vec(), b.to_vec());
    let mut out = vec![0u64; T::N];
    check(unsafe { op(fa.as_ptr(), fb.as_ptr(), out.as_mut_ptr(), 1) });
    T::read(&out)
}
fn un<T: Limbs>(op: Un, a: &T) -> T {
    init();
    let fa = a.to_vec();
    let mut out = vec![0u64; T::N];
    check(unsafe { op(fa.as_ptr(), out.as_mut_ptr(), 1) });
    T::read(&out)
}
/// n elements, one launch
pub fn bin_batch<T: Limbs>(op: Bin, a: &[T], b: &[T]) -> Vec<T> {
    init();
    assert_eq!(a.len(), b.len());
    let fa: Vec<u64> = a.iter().flat_map(|x| x.to_vec()).collect();
    let fb: Vec<u64> = b.iter().flat_map(|x| x.to_vec()).collect();
    let mut out = vec![0u64; fa.len()];
    check(unsafe { op(fa.as_ptr(), fb.as_ptr(), out.as_mut_ptr(), a.len()) });
    out.chunks(T::N).map(T::read).collect()
}

macro_rules! tower_ops {
    ($t:ident, $add:ident, $sub:ident, $mul:ident, $inv:ident, $neg:ident) => {
        impl $t {
            pub fn inv(&self) -> Self { un(ffi::$inv, self) }
            pub fn sq(&self) -> Self { bin(ffi::$mul, self, self) } // sq = self * self (fq2.rs:34-36)
            pub fn mul_batch(a: &[Self], b: &[Self]) -> Vec<Self> { bin_batch(ffi::$mul, a, b) }
        }
        impl<'a> Add<&'a $t> for &'a $t { type Output = $t; fn add(self, r: &$t) -> $t { bin(ffi::$add, self, r) } }
        impl<'a> Sub<&'a $t> for &'a $t { type Output = $t; fn sub(self, r: &$t) -> $t { bin(ffi::$sub, self, r) } }
        impl<'a> Mul<&'a $t> for &'a $t { type Output = $t; fn mul(self, r: &$t) -> $t { bin(ffi::$mul, self, r) } }
        impl Add<$t> for $t { type Output = $t; fn add(self, r: $t) -> $t { bin(ffi::$add, &self, &r) } }
        impl Sub<$t> for $t { type Output = $t; fn sub(self, r: $t) -> $t { bin(ffi::$sub, &self, &r) } }
        impl Mul<$t> for $t { type Output = $t; fn mul(self, r: $t) -> $t { bin(ffi::$mul, &self, &r) } }
        impl Neg for $t { type Output = $t; fn neg(self) -> $t { un(ffi::$neg, &self) } }
        impl<'a> Neg for &'a $t { type Output = $t; fn neg(self) -> $t { un(ffi::$neg, self) } }
    };
}
tower_ops!(Fq2, zkt_fq2_add_batch, zkt_fq2_sub_batch, zkt_fq2_mul_batch, zkt_fq2_inv_batch, zkt_fq2_neg_batch);
tower_ops!(Fq6, zkt_fq6_add_batch, zkt_fq6_sub_batch, zkt_fq6_mul_batch, zkt_fq6_inv_batch, zkt_fq6_neg_batch);
tower_ops!(Fq12, zkt_fq12_add_batch, zkt_fq12_sub_batch, zkt_fq12_mul_batch, zkt_fq12_inv_batch, zkt_fq12_neg_batch);

impl Fq2 {
    pub fn new(u1: &Fq1, u0: &Fq1) -> Self { Fq2 { u1: u1.clone(), u0: u0.clone() } } // fq2.rs:22-24
    pub fn reduce(&self) -> Self { un(ffi::zkt_fq2_reduce_batch, self) } // x (1+u), fq2.rs:52-58
}
impl Fq6 {
    pub fn new(v2: &Fq2, v1: &Fq2, v0: &Fq2) -> Self { Fq6 { v2: v2.clone(), v1: v1.clone(), v0: v0.clone() } }
    pub fn reduce(&self) -> Self { un(ffi::zkt_fq6_reduce_batch, self) } // x v, fq6.rs:54-62
}
impl Fq12 {
    pub fn new(w1: &Fq6, w0: &Fq6) -> Self { Fq12 { w1: w1.clone(), w0: w0.clone() } } // fq12.rs:24-29
    /// fq12.rs:42-57
    pub fn pow(&self, exp: &BigUint) -> Fq12 {
        init();
        let mut e = exp.to_u32_digits();
        if e.is_empty() { e.push(0); }
        let fa = self.to_vec();
        let mut out = vec![0u64; 72];
        check(unsafe { ffi::zkt_fq12_pow_batch(fa.as_ptr(), e.as_ptr(), e.len(), out.as_mut_ptr(), 1) });
        Fq12::read(&out)
    }
    /// `From<&dyn ToBigUint>` (fq12.rs:60-67): the scalar sits in w0.v0.u0
    pub fn from_scalar(n: &dyn ToBigUint) -> Fq12 {
        let z = Fq1::new(&0u8);
        let z2 = Fq2::new(&z, &z);
        let z6 = Fq6::new(&z2, &z2, &z2);
        let s = Fq1::new(&n.to_biguint().unwrap());
        Fq12::new(&z6, &Fq6::new(&z2, &z2, &Fq2::new(&z, &s)))
    }
}
