//! Pairing (pairing.rs:15-100) and GTPoint (gt_point.rs:4-39).
use crate::ffi;
use crate::points::{G1Point, G2Point};
use crate::tower::{Fq12, Limbs};
use crate::{check, check_bool, init};
use std::ops::Mul;

/// gt_point.rs:4-6 — the field `e` is private there too
#[derive(Clone, Debug)]
pub struct GTPoint { e: Fq12 }
impl GTPoint {
    pub fn new(e: &Fq12) -> Self { GTPoint { e: e.clone() } }
    pub(crate) fn e_clone(&self) -> Fq12 { self.e.clone() }
}
impl PartialEq for GTPoint { // gt_point.rs:33-39: all twelve coefficients
    fn eq(&self, other: &Self) -> bool {
        let (a, b) = (self.e.to_vec(), other.e.to_vec());
        check_bool(unsafe { ffi::zkt_gt_eq(a.as_ptr(), b.as_ptr()) })
    }
}
impl Eq for GTPoint {}
impl<'a> Mul<&'a GTPoint> for &'a GTPoint { type Output = GTPoint; fn mul(self, r: &GTPoint) -> GTPoint { GTPoint { e: &self.e * &r.e } } } // :16-31
impl Mul<GTPoint> for GTPoint { type Output = GTPoint; fn mul(self, r: GTPoint) -> GTPoint { GTPoint { e: &self.e * &r.e } } }

#[derive(Clone)]
pub struct Pairing; // pairing.rs:16-18: the bits of r-1 live in the kernels

type PairFn = unsafe extern "C" fn(*const ffi::zkt_g1_affine, *const ffi::zkt_g2_affine, *mut u64, usize) -> i32;
fn run(f: PairFn, p: &[G1Point], q: &[G2Point]) -> Vec<Fq12> {
    init();
    assert_eq!(p.len(), q.len());
    let a: Vec<ffi::zkt_g1_affine> = p.iter().map(|x| x.to_raw()).collect();
    let b: Vec<ffi::zkt_g2_affine> = q.iter().map(|x| x.to_raw()).collect();
    let mut out = vec![0u64; 72 * a.len()];
    check(unsafe { f(a.as_ptr(), b.as_ptr(), out.as_mut_ptr(), a.len()) }); // infinity -> the reference's panic (rational_function.rs:36,59)
    out.chunks(72).map(Fq12::read).collect()
}
unsafe extern "C" fn miller_g2g1(p: *const ffi::zkt_g1_affine, q: *const ffi::zkt_g2_affine, out: *mut u64, n: usize) -> i32 { ffi::zkt_miller_g2g1_batch(q, p, out, n) }

impl Pairing {
    pub fn new() -> Self { init(); Pairing } // pairing.rs:58-73
    pub fn calc_g1_g2(&self, p: &G1Point, q: &G2Point) -> Fq12 { run(ffi::zkt_miller_g1g2_batch, &[p.clone()], &[q.clone()]).remove(0) } // :54
    pub fn calc_g2_g1(&self, p: &G2Point, q: &G1Point) -> Fq12 { run(miller_g2g1, &[q.clone()], &[p.clone()]).remove(0) } // :55
    pub fn weil(&self, p1: &G1Point, p2: &G2Point) -> GTPoint { GTPoint::new(&run(ffi::zkt_weil_batch, &[p1.clone()], &[p2.clone()]).remove(0)) } // :75-84
    pub fn tate(&self, p1: &G1Point, p2: &G2Point) -> GTPoint { GTPoint::new(&run(ffi::zkt_tate_batch, &[p1.clone()], &[p2.clone()]).remove(0)) } // :86-100
    /// n independent pairings in one launch — what the GPU is for (BASELINE config 3: 2^16 of them)
    pub fn tate_batch(&self, p1: &[G1Point], p2: &[G2Point]) -> Vec<GTPoint> { run(ffi::zkt_tate_batch, p1, p2).iter().map(GTPoint::new).collect() }
}
