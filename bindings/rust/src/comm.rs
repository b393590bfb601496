//! Multi-GPU: one process per GPU; the only exchange is an all-gather of fixed-size Jacobian partial sums (include/zkt.h, "multi-GPU").
use crate::ffi;
use crate::polynomial::G1Bases;
use crate::points::G1Point;
use crate::{check, init};

/// rank 0: the RCCL unique id to ship to the other ranks (as with ncclGetUniqueId)
pub fn unique_id() -> [u8; ffi::ZKT_COMM_ID_BYTES] {
    init();
    let mut id = [0u8; ffi::ZKT_COMM_ID_BYTES];
    check(unsafe { ffi::zkt_comm_unique_id(id.as_mut_ptr()) });
    id
}
pub fn comm_init(rank: i32, world: i32, id: &[u8; ffi::ZKT_COMM_ID_BYTES]) { init(); check(unsafe { ffi::zkt_comm_init(rank, world, id.as_ptr()) }); }
pub fn comm_finalize() { unsafe { ffi::zkt_comm_finalize() } }
/// contiguous index range of `rank`
pub fn shard_range(n: usize, rank: i32, world: i32) -> (usize, usize) {
    let (mut lo, mut hi) = (0usize, 0usize);
    unsafe { ffi::zkt_comm_shard_range(n, rank, world, &mut lo, &mut hi) };
    (lo, hi)
}
/// eval_with_g1_hidings with the terms partitioned over the ranks: `bases` is this rank's shard; collective
pub unsafe fn eval_with_g1_hidings_sharded(bases: &G1Bases, dev_scalars: *const u64, stream: *mut std::os::raw::c_void) -> G1Point {
    let mut out = G1Point::zero_raw();
    check(ffi::zkt_g1_msm_sharded(bases.raw(), dev_scalars, bases.len(), stream, &mut out));
    G1Point::from_raw(&out)
}
