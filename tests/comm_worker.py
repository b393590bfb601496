"""One rank of the world > 1 rehearsal of the C ABI's multi-GPU entry points (include/zkt.h, "multi-GPU"): launched by
tests/test_gpu_comm.py, `world` processes sharing ONE card.  The exchange step goes through zkt_comm_init_callback with a gloo
all_gather (RCCL refuses two ranks on one device; on a real node bench.py uses zkt_comm_init = RCCL).  Everything else — the resident
shard, its Jacobian partial, the combine — is the code path of an 8-GPU run.  Prints `COMM_WORKER_OK <rank>` on success."""
import ctypes, importlib, os, sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zkt_testlib import oracle, ptr, SplitMix64, R, G1W, G2W, ints_to_arr, limbs_to_int, SECP_N
    from qap_util import chain_circuit_sparse, sparse_struct, alloc_crs
    zk = importlib.import_module("zk-toolkit_amd")
    zk.init(0)
    L, O = zk.lib(), oracle()

    CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)

    def allgather(_ctx, send, recv, nbytes):
        try:
            mine = torch.frombuffer((ctypes.c_uint8 * nbytes).from_address(send), dtype=torch.uint8).clone()
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, mine)
            buf = torch.cat(parts).numpy()                       # keep the temporary alive across the copy
            ctypes.memmove(recv, buf.ctypes.data, nbytes * world)
            return 0
        except Exception as e:       # never let an exception cross the C boundary
            print("allgather callback failed:", repr(e), flush=True)
            return 1
    cb = CB(allgather)
    assert L.zkt_comm_rank() == -1 and L.zkt_comm_world() == 0
    zk.check(L.zkt_comm_init_callback(rank, world, ctypes.cast(cb, ctypes.c_void_p), None))
    assert L.zkt_comm_rank() == rank and L.zkt_comm_world() == world
    assert L.zkt_comm_init_callback(rank, world, ctypes.cast(cb, ctypes.c_void_p), None) == zk.ZKT_ERR_SHAPE        # already initialised

    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
    # --- MSM sharded by index range, all three groups: equal to the whole MSM (one-shot entry point) on every rank
    for name, W, order, gen_fn in (("g1", G1W, R, "zkto_g1_generator"), ("g2", G2W, R, "zkto_g2_generator"), ("secp", 9, SECP_N, "zkto_secp_generator")):
        n = 2500 + 7 * (name == "g2")
        rng = np.random.Generator(np.random.PCG64(4242))          # every rank builds the same global problem
        ks = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64); ks[:, 3] >>= np.uint64(2)
        ss = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64); ss[:, 3] >>= np.uint64(2)
        g = np.zeros((1, W), np.uint64); getattr(O, gen_fn)(ptr(g))
        bases = np.zeros((n, W), np.uint64)
        zk.check(getattr(L, f"zkt_{name}_mul_batch")(ptr(np.repeat(g, n, axis=0)), ptr(ks), 4, ptr(bases), n))
        whole = np.zeros((1, W), np.uint64); zk.check(getattr(L, f"zkt_{name}_msm")(ptr(bases), ptr(ss), n, ptr(whole)))
        L.zkt_comm_shard_range(ctypes.c_size_t(n), rank, world, ctypes.byref(lo), ctypes.byref(hi))
        a, b = lo.value, hi.value
        h = ctypes.c_void_p(); zk.check(getattr(L, f"zkt_{name}_bases_upload")(ptr(bases[a:b].copy()), b - a, ctypes.byref(h)))
        d_s = torch.from_numpy(ss[a:b].copy().view(np.int64)).cuda()
        got = np.zeros((1, W), np.uint64)
        zk.check(getattr(L, f"zkt_{name}_msm_sharded")(h, vp(d_s), ctypes.c_size_t(b - a), None, ptr(got)))
        assert (got == whole).all(), f"{name}: sharded MSM differs from the whole MSM on rank {rank}"
        # pipelined form: two MSMs in flight, collected through the exchange
        got2 = np.zeros((2, W), np.uint64)
        for slot in (0, 1): zk.check(getattr(L, f"zkt_{name}_msm_submit")(h, vp(d_s), b - a, None, slot))
        for slot in (0, 1): zk.check(getattr(L, f"zkt_{name}_msm_sharded_collect")(h, slot, ptr(got2[slot:slot + 1])))
        assert (got2 == whole).all(), name
        getattr(L, f"zkt_{name}_bases_free")(h)
        tot = sum(limbs_to_int(x) * limbs_to_int(y) for x, y in zip(ks, ss)) % order       # and to the oracle, by linearity
        want = np.zeros((1, W), np.uint64)
        assert getattr(O, f"zkto_{name}_mul_batch")(ptr(g), ptr(ints_to_arr([tot], 4)), 4, ptr(want), 1, 1) == 0
        assert (got == want).all(), name

    # --- one Groth16 proof sharded over the ranks (BASELINE config 4) == the unsharded proof
    n = 300
    mats, wires, l, m = chain_circuit_sparse(n, seed=5)
    rng = SplitMix64(999)
    fr = lambda x: ints_to_arr([x], 4)
    trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
    r, s = fr(rng.below(R - 1) + 1), fr(rng.below(R - 1) + 1)
    structs = [sparse_struct(*M) for M in mats]
    L.zkt_groth16_prove_r1cs_sharded.argtypes = [ctypes.c_void_p] * 7
    d_w = torch.from_numpy(wires.view(np.int64)).cuda()
    vk, vbuf = alloc_crs(1, l, m); pk = ctypes.c_void_p()
    zk.check(L.zkt_groth16_setup_r1cs(n, l, m, *[ctypes.addressof(x) for x in structs], *[t.ctypes.data for t in trap], ctypes.addressof(vk), ctypes.addressof(pk)))
    want = (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
    zk.check(L.zkt_groth16_prove_r1cs_dev(pk, d_w.data_ptr(), r.ctypes.data, s.ctypes.data, *[x.ctypes.data for x in want]))
    L.zkt_groth16_pk_free(pk)
    vk2, vbuf2 = alloc_crs(1, l, m); pk2 = ctypes.c_void_p()
    zk.check(L.zkt_groth16_setup_r1cs_sharded(n, l, m, *[ctypes.addressof(x) for x in structs], *[t.ctypes.data for t in trap], rank, world,
                                               ctypes.addressof(vk2), ctypes.addressof(pk2)))
    got = (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
    zk.check(L.zkt_groth16_prove_r1cs_sharded(pk2, d_w.data_ptr(), r.ctypes.data, s.ctypes.data, *[x.ctypes.data for x in got]))
    for x, y, nm in zip(want, got, "ABC"):
        assert (x == y).all(), f"sharded proof element {nm} differs on rank {rank}"
    L.zkt_groth16_pk_free(pk2)
    L.zkt_comm_finalize()
    assert L.zkt_comm_world() == 0
    dist.barrier()
    dist.destroy_process_group()
    print(f"COMM_WORKER_OK {rank}", flush=True)


if __name__ == "__main__":
    main()
