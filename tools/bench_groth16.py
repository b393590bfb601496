#!/usr/bin/env python3
"""Groth16 prove + verify at BASELINE config 4 scale (synthetic chain R1CS, n constraints) through
zkt_groth16_setup_r1cs / zkt_groth16_prove_r1cs.  Prints setup time, proofs/s and the verifier's decision."""
import argparse, ctypes, importlib, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from zkt_testlib import R, SplitMix64, ints_to_arr, ptr, G1W, G2W
from qap_util import chain_circuit_sparse, sparse_struct, alloc_crs

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--proofs", type=int, default=5)
ap.add_argument("--rank", type=int, default=-1, help="with --shard-of: time this rank only, and nothing else in the process (no unsharded key before it)")
ap.add_argument("--shard-of", type=int, default=0, help="W: also time one rank's share of a proof sharded over W GPUs (ranks 0, W/2, W-1, one after the other on this card)")
args = ap.parse_args()
n = 1 << args.log_n
zk = importlib.import_module("zk-toolkit_amd"); zk.init(); L = zk.lib()
t0 = time.perf_counter()
mats, wires, l, m = chain_circuit_sparse(n, seed=7)
print(f"R1CS n={n} m={m} built in {time.perf_counter()-t0:.1f}s (host, python)", flush=True)
rng = SplitMix64(7)
fr = lambda v: ints_to_arr([v], 4)
trap = [fr(rng.below(R - 1) + 1) for _ in range(5)]
r, s = fr(rng.below(R - 1) + 1), fr(rng.below(R - 1) + 1)
structs = [sparse_struct(*M) for M in mats]
vk, vbuf = alloc_crs(1, l, m)
vk.g1_uvw_wit = None
import torch
d_wires = torch.from_numpy(wires.view(np.int64)).cuda(); torch.cuda.synchronize()
ok = 1
if args.rank < 0:
    pk = ctypes.c_void_p()
    t0 = time.perf_counter()
    zk.check(L.zkt_groth16_setup_r1cs(n, l, m, *[ctypes.addressof(x) for x in structs], *[t.ctypes.data for t in trap], ctypes.addressof(vk), ctypes.addressof(pk)))
    print(f"setup (CRS::new + resident MSM tables): {time.perf_counter()-t0:.2f}s", flush=True)
    gp = (np.zeros((1, G1W), np.uint64), np.zeros((1, G2W), np.uint64), np.zeros((1, G1W), np.uint64))
    zk.check(L.zkt_groth16_prove_r1cs(pk, wires.ctypes.data, r.ctypes.data, s.ctypes.data, *[x.ctypes.data for x in gp]))   # warm-up (workspaces)
    host = gp[0].copy(), gp[1].copy(), gp[2].copy()
    t0 = time.perf_counter()
    for _ in range(args.proofs):
        zk.check(L.zkt_groth16_prove_r1cs_dev(pk, d_wires.data_ptr(), r.ctypes.data, s.ctypes.data, *[x.ctypes.data for x in gp]))
    dt = (time.perf_counter() - t0) / args.proofs
    assert all((a == b).all() for a, b in zip(host, gp)), "device-wires proof differs from host-wires proof"
    stmt = wires[:l + 1].copy()
    t0 = time.perf_counter()
    ok = L.zkt_groth16_verify(ctypes.byref(vk), ptr(gp[0]), ptr(gp[1]), ptr(gp[2]), ptr(stmt), l + 1)
    tv = time.perf_counter() - t0
    print(f"prove: {dt*1e3:.1f} ms/proof = {1/dt:.2f} proofs/s (wires resident in HBM); verify -> {ok} in {tv*1e3:.1f} ms")
    L.zkt_groth16_pk_free(pk)
if args.shard_of > 1:
    # what ONE rank of a W-GPU proof does before the 672-byte exchange: its index ranges of the three base sets, its own range of the quotient
    W = args.shard_of
    parts = torch.zeros(1024, dtype=torch.int32, device="cuda")
    for rank in (sorted({0, W // 2, W - 1}) if args.rank < 0 else [args.rank]):
        spk = ctypes.c_void_p()
        zk.check(L.zkt_groth16_setup_r1cs_sharded(n, l, m, *[ctypes.addressof(x) for x in structs], *[t.ctypes.data for t in trap], rank, W, ctypes.addressof(vk), ctypes.addressof(spk)))
        for _ in range(2): zk.check(L.zkt_groth16_prove_r1cs_partials(spk, d_wires.data_ptr(), r.ctypes.data, s.ctypes.data, parts.data_ptr()))
        torch.cuda.synchronize(); ts = []
        for _ in range(args.proofs):
            t0 = time.perf_counter(); zk.check(L.zkt_groth16_prove_r1cs_partials(spk, d_wires.data_ptr(), r.ctypes.data, s.ctypes.data, parts.data_ptr())); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f"rank {rank} of {W}: its share of one proof {min(ts)*1e3:.2f} ms best, {sorted(ts)[len(ts)//2]*1e3:.2f} ms median (one at a time; the exchange is 672 B more)", flush=True)
        L.zkt_groth16_pk_free(spk)
sys.exit(0 if ok == 1 else 1)
