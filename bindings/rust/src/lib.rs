//! Drop-in host side for zk-toolkit's data-parallel hot path on MI355X.
//!
//! Every type keeps the reference's name and shape (paths relative to the reference crate's `src/`):
//!   `Fq1`, `Fr`            building_block/field/prime_field_elem.rs:263-457 (PrimeFieldElem over q / r)
//!   `Fq2`, `Fq6`, `Fq12`   building_block/curves/bls12_381/{fq2,fq6,fq12}.rs
//!   `G1Point`, `G2Point`   building_block/curves/bls12_381/{g1_point,g2_point}.rs, curves/macros.rs
//!   `GTPoint`, `Pairing`   building_block/curves/bls12_381/{gt_point,pairing}.rs
//!   `Polynomial`           building_block/field/polynomial.rs:271-293 (eval_with_g1_hidings / eval_with_g2_hidings)
//!   `groth16::*`           zk/w_trusted_setup/groth16/zktoolkit_based/{crs,prover,verifier,proof}.rs
//!   `Bulletproofs`         zk/wo_trusted_setup/bulletproofs.rs
//!   `pinocchio::*`         zk/w_trusted_setup/pinocchio/{crs,prover,verifier,proof,witness}.rs
//!   `Signer`, `PrivateKey` building_block/curves/bls12_381/{signature,private_key}.rs
//! and forwards to the batch-first C ABI (`ffi`, generated from include/zkt.h).  Conventions carried over from the only native
//! backend the reference has (building_block/mcl/): one global `init` behind `Once` that panics on failure (mcl_initializer.rs:4-15),
//! out-parameter free functions underneath, value types with operator overloads on top.  A non-OK status becomes `panic!`, because
//! the reference panics in the same places (inverse of zero, pairing with the point at infinity, index mismatch).
//!
//! Single-element operators (`&a * &b`) are legal and bit-identical but launch one GPU kernel each; the `*_batch` associated functions
//! are what a hot loop should call.  There is NO CPU arithmetic in this crate: values are marshalled to limbs and handed to the library.
pub mod ffi;
pub mod field;
pub mod tower;
pub mod points;
pub mod pairing;
pub mod polynomial;
pub mod groth16;
pub mod bulletproofs;
pub mod signature;
pub mod pinocchio;
pub mod comm;
pub mod reference_paths;

pub use field::{Fq1, Fr, SecpFq, SecpFr, PrimeField, PrimeFieldElem, PrimeFieldElems, SparseVec};
pub use tower::{Fq2, Fq6, Fq12};
pub use points::{AffinePoint, AffinePoints, G1Point, G2Point, SecpPoint};
pub use pairing::{GTPoint, Pairing};
pub use polynomial::{Polynomial, G1Bases, G2Bases};
pub use bulletproofs::Bulletproofs;
pub use signature::{PrivateKey, Signer};
/// the reference crate's own module paths (`building_block::curves::bls12_381::g1_point::G1Point`, `zk::w_trusted_setup::groth16::zktoolkit_based::prover::Prover`, ...)
/// and its runtime-order `PrimeField` / `PrimeFieldElem`: reference_paths.rs
pub use reference_paths::{building_block, zk};

use std::ffi::CStr;
use std::sync::Once;

static INIT: Once = Once::new();

/// `MclInitializer::init()` of the reference's native backend (mcl_initializer.rs:4-15): call once, anywhere; panics if no MI355X is usable.
/// Every constructor of this crate calls it, so explicit use is optional.  `ZKT_DEVICE` selects the HIP ordinal (default: current device).
pub fn init() {
    INIT.call_once(|| {
        let dev = std::env::var("ZKT_DEVICE").ok().and_then(|s| s.parse::<i32>().ok()).unwrap_or(-1);
        let rc = unsafe { ffi::zkt_init(dev) };
        if rc != ffi::ZKT_OK {
            panic!("Failed to initialize the MI355X engine: {}", strerror(rc));
        }
    });
}

pub(crate) fn strerror(rc: i32) -> String {
    unsafe { CStr::from_ptr(ffi::zkt_strerror(rc)).to_string_lossy().into_owned() }
}

/// status -> panic, with the reference's wording where it has one
pub(crate) fn check(rc: i32) {
    if rc == ffi::ZKT_OK {
        return;
    }
    let idx = unsafe { ffi::zkt_last_error_index() };
    match rc {
        ffi::ZKT_ERR_INV_ZERO => panic!("Cannot find inverse of zero (element {})", idx), // prime_field_elem.rs:380-382
        ffi::ZKT_ERR_INFINITY => panic!("Both points need to be rational (element {})", idx), // rational_function.rs:36
        _ => panic!("zkt: {}", strerror(rc)),
    }
}

/// bool-returning entry points: 1 / 0, negative = -status
pub(crate) fn check_bool(rc: i32) -> bool {
    if rc < 0 {
        check(-rc);
    }
    rc == 1
}
