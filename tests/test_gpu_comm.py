"""Multi-GPU entry points of the C ABI (include/zkt.h "multi-GPU", csrc/zkt_comm.cpp).
 * world = 1 through RCCL's own code path (zkt_comm_unique_id / zkt_comm_init): the sharded calls equal the unsharded ones;
 * world = 2 with two processes on this one card and the callback transport over gloo (tests/comm_worker.py): resident shards,
   Jacobian partials, the exchange and the combine are the code of an 8-GPU run — only the wire differs."""
import ctypes, importlib, os, subprocess, sys
import numpy as np
import pytest
from zkt_testlib import *

pytestmark = pytest.mark.gpu
zk = importlib.import_module("zk-toolkit_amd")
O = oracle()
HERE = os.path.dirname(os.path.abspath(__file__))


def test_comm_world1_rccl_matches_unsharded():
    import torch
    zk.init()
    L = zk.lib()
    ident = (ctypes.c_uint8 * 128)()
    zk.check(L.zkt_comm_unique_id(ident))
    assert any(ident), "RCCL returned an all-zero unique id"
    assert L.zkt_g1_msm_sharded(None, None, ctypes.c_size_t(0), None, None) == ZKT_ERR_SHAPE          # before zkt_comm_init
    zk.check(L.zkt_comm_init(0, 1, ctypes.cast(ident, ctypes.c_void_p)))
    try:
        assert L.zkt_comm_rank() == 0 and L.zkt_comm_world() == 1
        n = 3000
        rng = np.random.Generator(np.random.PCG64(5))
        ks = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64); ks[:, 3] >>= np.uint64(2)
        ss = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64); ss[:, 3] >>= np.uint64(2)
        g = np.zeros((1, G1W), np.uint64); O.zkto_g1_generator(ptr(g))
        bases = np.zeros((n, G1W), np.uint64)
        zk.check(L.zkt_g1_mul_batch(ptr(np.repeat(g, n, axis=0)), ptr(ks), 4, ptr(bases), n))
        h = ctypes.c_void_p(); zk.check(L.zkt_g1_bases_upload(ptr(bases), n, ctypes.byref(h)))
        d_s = torch.from_numpy(ss.view(np.int64)).cuda()
        want, got = np.zeros((1, G1W), np.uint64), np.zeros((1, G1W), np.uint64)
        zk.check(L.zkt_g1_msm_dev(h, ctypes.c_void_p(d_s.data_ptr()), n, None, ptr(want), None))
        zk.check(L.zkt_g1_msm_sharded(h, ctypes.c_void_p(d_s.data_ptr()), ctypes.c_size_t(n), None, ptr(got)))
        assert (got == want).all()
        L.zkt_g1_bases_free(h)
        lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
        for nn in (0, 1, 7, 8, 1 << 20):
            for world in (1, 2, 3, 8):
                r = []
                for k in range(world):
                    L.zkt_comm_shard_range(ctypes.c_size_t(nn), k, world, ctypes.byref(lo), ctypes.byref(hi)); r.append((lo.value, hi.value))
                assert r[0][0] == 0 and r[-1][1] == nn and all(r[i][1] == r[i + 1][0] for i in range(world - 1))
                assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    finally:
        L.zkt_comm_finalize()


def test_collective_entry_points_take_part_in_the_exchange_when_the_local_stage_fails():
    """A rank whose local stage fails (here: a NULL handle / a NULL output) still sends its status word, so no other rank can be left inside
    ncclAllGather; every rank returns the error.  At world 1 this runs the real RCCL all-gather with a failing status."""
    import torch
    zk.init()
    L = zk.lib()
    zk.check(L.zkt_comm_init(0, 1, None))                                      # world 1: the library draws its own unique id
    try:
        out = np.zeros((1, G1W), np.uint64)
        assert L.zkt_g1_msm_sharded(None, None, ctypes.c_size_t(0), None, ptr(out)) == ZKT_ERR_SHAPE
        assert L.zkt_g1_msm_sharded_collect(None, 0, ptr(out)) == ZKT_ERR_SHAPE
        assert L.zkt_groth16_prove_r1cs_sharded(None, None, None, None, None, None, None) == ZKT_ERR_SHAPE
        # and the communicator is still usable afterwards
        n = 300
        rng = np.random.Generator(np.random.PCG64(11))
        ss = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64); ss[:, 3] >>= np.uint64(2)
        g = np.zeros((1, G1W), np.uint64); O.zkto_g1_generator(ptr(g))
        bases = np.zeros((n, G1W), np.uint64)
        zk.check(L.zkt_g1_mul_batch(ptr(np.repeat(g, n, axis=0)), ptr(ss), 4, ptr(bases), n))
        h = ctypes.c_void_p(); zk.check(L.zkt_g1_bases_upload(ptr(bases), n, ctypes.byref(h)))
        d_s = torch.from_numpy(ss.view(np.int64)).cuda()
        want, got = np.zeros((1, G1W), np.uint64), np.zeros((1, G1W), np.uint64)
        zk.check(L.zkt_g1_msm_dev(h, ctypes.c_void_p(d_s.data_ptr()), n, None, ptr(want), None))
        zk.check(L.zkt_g1_msm_sharded(h, ctypes.c_void_p(d_s.data_ptr()), ctypes.c_size_t(n), None, ptr(got)))
        assert (got == want).all()
        L.zkt_g1_bases_free(h)
    finally:
        L.zkt_comm_finalize()


def test_shutdown_then_init_rebuilds_device_bound_state():
    """zkt_shutdown releases the generator comb tables, the guard side stream and the communicator (they live on the device of that zkt_init);
    after a second zkt_init the entry points that use them give the same bits (they used to survive as dangling device state)."""
    zk.init()
    L = zk.lib()
    sk = ints_to_arr([5, 77, R - 1], 4)
    pk1 = np.zeros((3, G1W), np.uint64); zk.check(L.zkt_bls_public_keys_batch(ptr(sk), 3, ptr(pk1)))       # builds the G1 comb table
    g1 = np.zeros((1, G1W), np.uint64); O.zkto_g1_generator(ptr(g1))
    g2 = np.zeros((1, G2W), np.uint64); O.zkto_g2_generator(ptr(g2))
    e1 = np.zeros((1, FQ12), np.uint64); zk.check(L.zkt_tate_batch(ptr(g1), ptr(g2), ptr(e1), 1))          # small batch: guard side stream
    zk.check(L.zkt_comm_init(0, 1, None))
    L.zkt_shutdown()
    assert L.zkt_comm_world() == 0, "zkt_shutdown finalises the communicator"
    assert L.zkt_tate_batch(ptr(g1), ptr(g2), ptr(e1), 1) == ZKT_ERR_DEVICE
    zk.init()
    pk2 = np.zeros((3, G1W), np.uint64); zk.check(L.zkt_bls_public_keys_batch(ptr(sk), 3, ptr(pk2)))
    e2 = np.zeros((1, FQ12), np.uint64); zk.check(L.zkt_tate_batch(ptr(g1), ptr(g2), ptr(e2), 1))
    assert (pk1 == pk2).all() and (e1 == e2).all()
    want = np.zeros((3, G1W), np.uint64)
    assert O.zkto_g1_mul_batch(ptr(np.repeat(g1, 3, axis=0)), ptr(sk), 4, ptr(want), 3, 1) == 0
    assert (pk2 == want).all()
    zk.check(L.zkt_comm_init(0, 1, None)); L.zkt_comm_finalize()


def test_comm_world2_callback_transport_two_processes_one_card():
    world, port = 2, 29700 + (os.getpid() % 200)
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "comm_worker.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs: q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"COMM_WORKER_OK {rank}" in out, f"rank {rank} rc={p.returncode}\n{out[-3000:]}"
