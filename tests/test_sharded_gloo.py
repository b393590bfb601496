"""N>1 path on CPU: world_size 2 over gloo.  The sharding + all_gather + local-combine orchestration of
tests/sharded.py (test-only orchestration helper; the product's exchange lives in csrc/zkt_comm.cpp) is run with the oracle injected as the compute (no GPU here) and must equal the
unsharded MSM (polynomial.rs:271-281).  On GPUs bench.py runs the same orchestration with the HIP kernels."""
import importlib, os, sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zkt_testlib import oracle, ptr, SplitMix64, R, G1W, ints_to_arr
    import sharded as sh
    O = oracle()
    rng = SplitMix64(99)                                   # every rank builds the same global problem
    g = np.zeros((1, G1W), dtype=np.uint64); O.zkto_g1_generator(ptr(g))
    bases = np.zeros((n, G1W), dtype=np.uint64)
    assert O.zkto_g1_mul_batch(ptr(np.repeat(g, n, axis=0)), ptr(ints_to_arr([rng.below(R) for _ in range(n)], 4)), 4, ptr(bases), n, 2) == 0
    scalars = ints_to_arr([rng.below(R) for _ in range(n)], 4)
    lo, hi = sh.shard_range(n, rank, world)
    part = np.zeros((1, G1W), dtype=np.uint64)
    assert O.zkto_g1_msm(ptr(bases[lo:hi].copy()), ptr(scalars[lo:hi].copy()), 4, hi - lo, ptr(part)) == 0

    def combine(stack):
        acc = np.zeros((1, G1W), dtype=np.uint64); acc[0, 12] = 1
        for row in stack.numpy().view(np.uint64):
            nxt = np.zeros_like(acc); O.zkto_g1_add_batch(ptr(acc), ptr(row.reshape(1, G1W).copy()), ptr(nxt), 1); acc = nxt
        return acc
    total = sh.sharded_sum(torch.from_numpy(part.view(np.int64)[0].copy()), combine)
    if rank == 0:
        want = np.zeros((1, G1W), dtype=np.uint64)
        assert O.zkto_g1_msm(ptr(bases), ptr(scalars), 4, n, ptr(want)) == 0
        q.put(bool((total == want).all()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_partition():
    import sharded as sh
    for n in (0, 1, 7, 8, 1 << 20):
        for world in (1, 2, 3, 8):
            r = [sh.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_sharded_msm_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, q)) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join(timeout=240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) is True
