//! BLS signatures — building_block/curves/bls12_381/signature.rs:8-40, private_key.rs:10-27, G2Point::hash_to_g2point g2_point.rs:84-88.
use crate::ffi;
use crate::field::Fr;
use crate::points::{G1Point, G2Point};
use crate::{check, init};

/// private_key.rs:10-12 — the reference draws `value` in [1, r-1] from OS entropy; here the caller supplies it
pub struct PrivateKey { pub value: Fr }

pub struct Signer { g1: G1Point } // signature.rs:8-11

fn offsets(msgs: &[Vec<u8>]) -> (Vec<u8>, Vec<u64>) {
    let mut flat = Vec::new();
    let mut off = vec![0u64];
    for m in msgs { flat.extend_from_slice(m); off.push(flat.len() as u64); }
    (flat, off)
}

impl Signer {
    pub fn new() -> Self { init(); Signer { g1: G1Point::g() } } // :14-22
    /// G1 generator * sk (:24-27)
    pub fn gen_public_key(&self, sk: &PrivateKey) -> G1Point { self.gen_public_keys(&[sk.value.clone()]).remove(0) }
    /// n keys, one launch through the generator's comb table
    pub fn gen_public_keys(&self, sks: &[Fr]) -> Vec<G1Point> {
        let k = Fr::flatten(sks);
        let mut out = vec![G1Point::zero_raw(); sks.len()];
        check(unsafe { ffi::zkt_bls_public_keys_batch(k.as_ptr(), sks.len(), out.as_mut_ptr()) });
        out.iter().map(G1Point::from_raw).collect()
    }
    /// hash_to_g2point(m) * sk (:28-31)
    pub fn sign(&self, m: &Vec<u8>, sk: &PrivateKey) -> G2Point { self.sign_batch(&[m.clone()], &[sk.value.clone()]).remove(0) }
    /// tate(g1, sig) == tate(pk, hash_to_g2point(m)) (:34-39)
    pub fn verify(&self, m: &Vec<u8>, sig: &G2Point, pk: &G1Point) -> bool { self.verify_batch(&[m.clone()], &[sig.clone()], &[pk.clone()])[0] }

    /// n messages and keys, one launch
    pub fn sign_batch(&self, msgs: &[Vec<u8>], sks: &[Fr]) -> Vec<G2Point> {
        assert_eq!(msgs.len(), sks.len());
        let (flat, off) = offsets(msgs);
        let k = Fr::flatten(sks);
        let mut out = vec![G2Point::zero_raw(); msgs.len()];
        check(unsafe { ffi::zkt_bls_sign_batch(flat.as_ptr(), off.as_ptr(), k.as_ptr(), msgs.len(), out.as_mut_ptr()) });
        out.iter().map(G2Point::from_raw).collect()
    }
    /// one signature per lane group: two Miller loops side by side, one final exponentiation
    pub fn verify_batch(&self, msgs: &[Vec<u8>], sigs: &[G2Point], pks: &[G1Point]) -> Vec<bool> {
        assert!(msgs.len() == sigs.len() && msgs.len() == pks.len());
        let (flat, off) = offsets(msgs);
        let s: Vec<ffi::zkt_g2_affine> = sigs.iter().map(|x| x.to_raw()).collect();
        let p: Vec<ffi::zkt_g1_affine> = pks.iter().map(|x| x.to_raw()).collect();
        let mut ok = vec![0u32; msgs.len()];
        check(unsafe { ffi::zkt_bls_verify_batch(flat.as_ptr(), off.as_ptr(), s.as_ptr(), p.as_ptr(), msgs.len(), ok.as_mut_ptr()) });
        ok.iter().map(|v| *v == 1).collect()
    }
}
/// G2Point::hash_to_g2point (g2_point.rs:84-88): the bytes as a big-endian integer mod r, times the G2 generator
pub fn hash_to_g2point(m: &Vec<u8>) -> G2Point {
    init();
    let off = [0u64, m.len() as u64];
    let mut out = G2Point::zero_raw();
    check(unsafe { ffi::zkt_bls_hash_to_g2_batch(m.as_ptr(), off.as_ptr(), 1, &mut out) });
    G2Point::from_raw(&out)
}
