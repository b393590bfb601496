#!/usr/bin/env python3
"""bench.py — G1 MSM throughput at 2^20 bases per GPU (BASELINE.json configs[1]), plus the Tate-pairing rate (configs[2]) and Groth16
proofs/s at 2^20 constraints (configs[3]) on MI355X.  Contract: python bench.py --gpus N --steps K --warmup W (N>1 under
torch.distributed.run, one rank per GPU).  One JSON line on rank 0.

step  = one MSM over 2^20 device-resident bases (a CRS) and 2^20 device-resident 255-bit scalars, result normalised to an affine
        point on the host.
N>1   = each rank owns a 2^20-term shard of one N*2^20-term MSM (weak scaling).  The exchange is inside the C ABI
        (zkt_comm_init + zkt_g1_msm_sharded_collect, csrc/zkt_comm.cpp): an RCCL all-gather of the 168-byte Jacobian partials over
        xGMI and an on-device combine.  torch.distributed only ships the RCCL unique id and does the barriers / the max over ranks.
value = total scalar-muls per second over all ranks (terms / wall time, max over ranks).
Beside it for N>1: `strong` (ONE 2^20-term MSM cut N ways) and `groth16` both sharded (one proof over N GPUs) and as replicas
(N keys, N independent proofs in flight) — the honest multi-GPU picture, not only the embarrassingly parallel one."""
import argparse, ctypes, importlib, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # the MSM pipeline wants its three stage streams on distinct hardware queues
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MSM_BYTES_PER_TERM = 128       # 96 B affine base + 32 B scalar
PAIRING_BYTES = 864            # 96 + 192 in, 576 out
PROFILE_ROUND = "r04"


def load_profile(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except Exception:
        return None


def rand_scalars_mod_r(seed, n):
    """n uniform scalars in [0, r) as (n,4) u64 — 255-bit draws with rejection (vectorised)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    r_limbs = [(R_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    out = np.zeros((0, 4), dtype=np.uint64)
    while out.shape[0] < n:
        c = rng.integers(0, 2**64, size=(int((n - out.shape[0]) * 1.2) + 16, 4), dtype=np.uint64)
        c[:, 3] &= np.uint64(0x7FFFFFFFFFFFFFFF)
        lt = np.zeros(c.shape[0], dtype=bool); eq = np.ones(c.shape[0], dtype=bool)
        for i in (3, 2, 1, 0):
            lt |= eq & (c[:, i] < np.uint64(r_limbs[i])); eq &= c[:, i] == np.uint64(r_limbs[i])
        out = np.concatenate([out, c[lt]])
    return np.ascontiguousarray(out[:n])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--pairings", type=int, default=1 << 16, help="pairings in the secondary measurement (0 = skip)")
    ap.add_argument("--groth16-log2n", type=int, default=20, help="constraints (log2) of the Groth16 prove+verify leg (0 = skip)")
    ap.add_argument("--groth16-proofs", type=int, default=8)
    ap.add_argument("--g2-log2n", type=int, default=20, help="terms (log2) of the resident G2 MSM leg: the B sum of a Groth16 proof (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-bulletproofs", dest="bulletproofs", action="store_false", help="skip the Bulletproofs leg (BASELINE config 5)")
    ap.add_argument("--scalar-dist", default="uniform", choices=["uniform", "ones", "bits"],
                    help="uniform in [0,r) (the metric) | all ones | random 0/1 (skew stress: one hot bucket)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    ndev = torch.cuda.device_count()
    backend = os.environ.get("ZKT_BENCH_BACKEND", "nccl")     # "gloo" only to rehearse the N>1 control flow on a 1-GPU box
    if backend == "nccl":
        assert local < ndev, f"LOCAL_RANK {local} but only {ndev} GPUs visible"
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    ctl_dev = dev if backend == "nccl" else "cpu"             # where the control-plane tensors (timings, flags, the RCCL id) live

    def all_ok(flag):
        """every rank leaves a leg together: True only if every rank's flag is True"""
        if world == 1: return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=ctl_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def max_over_ranks(x):
        if world == 1: return x
        t = torch.tensor([x], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    zk = importlib.import_module("zk-toolkit_amd")
    zk.init(local)
    L = zk.lib()
    stream = torch.cuda.current_stream()
    sp = ctypes.c_void_p(stream.cuda_stream)
    n = 1 << args.log2n
    vp = lambda t: ctypes.c_void_p(t.data_ptr())

    # ---- the exchange step lives in the C ABI: RCCL communicator of the library (or, for a one-GPU rehearsal, the callback transport over gloo)
    _cb_keep = []
    transport = "none"
    if world > 1:
        def init_callback_transport():
            """the library's host-callback transport, carried by torch.distributed (gloo tensors, or nccl through a device staging tensor)"""
            CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)

            def allgather(_ctx, send, recv, nbytes):
                try:
                    mine = torch.frombuffer((ctypes.c_uint8 * nbytes).from_address(send), dtype=torch.uint8).clone()
                    if backend == "nccl": mine = mine.to(dev)
                    parts = [torch.empty_like(mine) for _ in range(world)]
                    dist.all_gather(parts, mine)
                    buf = torch.cat(parts).cpu().numpy()
                    ctypes.memmove(recv, buf.ctypes.data, nbytes * world)
                    return 0
                except Exception as e:
                    print("allgather callback failed:", repr(e), file=sys.stderr, flush=True)
                    return 1
            cb = CB(allgather); _cb_keep.append(cb)
            zk.check(L.zkt_comm_init_callback(rank, world, ctypes.cast(cb, ctypes.c_void_p), None))

        if backend == "nccl":
            # RCCL communicator inside the library; the unique id travels over torch.distributed.  If any rank cannot create it, every rank
            # falls back to the callback transport together (the data path is the same, only the wire differs).
            ident = torch.zeros(128, dtype=torch.uint8)
            rc_id = 0
            if rank == 0:
                buf = (ctypes.c_uint8 * 128)()
                rc_id = L.zkt_comm_unique_id(buf)
                ident = torch.tensor(list(buf), dtype=torch.uint8)
            ident = ident.to(dev); dist.broadcast(ident, 0)
            idb = (ctypes.c_uint8 * 128)(*ident.cpu().tolist())
            rc_init = L.zkt_comm_init(rank, world, ctypes.cast(idb, ctypes.c_void_p)) if all_ok(rc_id == 0) else -1
            if all_ok(rc_init == 0):
                transport = "rccl (libzkt_hip.so's own communicator)"
            else:
                L.zkt_comm_finalize()
                init_callback_transport()
                transport = "callback over torch.distributed/nccl (RCCL communicator inside the library could not be created: rc %d)" % rc_init
        else:
            init_callback_transport()
            transport = "callback over torch.distributed/" + backend

    # ---- synthetic inputs, resident in HBM: P_i = k_i * G1 (seed 3 + rank), s_i uniform in [0,r) (seed 4 + rank)
    from zkt_testlib import G1_GEN, int_to_limbs, limbs_to_int, py_g1_mul, g1_arr
    gen = np.zeros((1, 13), dtype=np.uint64); gen[0, :6] = int_to_limbs(G1_GEN[0], 6); gen[0, 6:12] = int_to_limbs(G1_GEN[1], 6)
    d_gen = torch.from_numpy(np.repeat(gen, n, axis=0).view(np.int64)).to(dev)
    h_k = rand_scalars_mod_r(3 + 1000 * rank, n)
    d_k = torch.from_numpy(h_k.view(np.int64)).to(dev)
    d_bases = torch.empty((n, 13), dtype=torch.int64, device=dev)
    zk.check(L.zkt_g1_mul_batch_dev(vp(d_gen), vp(d_k), 4, vp(d_bases), n, sp))
    torch.cuda.synchronize()
    h = ctypes.c_void_p()
    t0 = time.perf_counter()
    zk.check(L.zkt_g1_bases_from_device(vp(d_bases), n, sp, ctypes.byref(h)))
    setup_s = time.perf_counter() - t0
    del d_gen
    h_scalars = rand_scalars_mod_r(4 + 1000 * rank, n)
    if args.scalar_dist == "ones":
        h_scalars[:] = 0; h_scalars[:, 0] = 1
    elif args.scalar_dist == "bits":
        h_scalars[:, 1:] = 0; h_scalars[:, 0] &= np.uint64(1)
    d_scalars = torch.from_numpy(h_scalars.view(np.int64)).to(dev)
    # a SECOND scalar set: the timed loop alternates between the two (step i takes set i % 2) and the last result of EACH is checked at full size, so no step can
    # have been served from anything the previous one left behind
    h_scalars_b = rand_scalars_mod_r(5 + 1000 * rank, n)
    if args.scalar_dist == "ones":
        h_scalars_b[:] = 0; h_scalars_b[:, 0] = 2
    elif args.scalar_dist == "bits":
        h_scalars_b[:, 1:] = 0; h_scalars_b[:, 0] &= np.uint64(1)
    d_scalars_b = torch.from_numpy(h_scalars_b.view(np.int64)).to(dev)
    out = np.zeros((1, 13), dtype=np.uint64)
    outp = out.ctypes.data_as(ctypes.c_void_p)
    last_of_set = [None, None]

    DEPTH = int(os.environ.get("ZKT_BENCH_DEPTH", "5"))   # MSMs in flight: sort / accumulate / reduce-tail of consecutive MSMs overlap
    NSLOT = 8          # ZKT_MSM_SLOTS

    def run(handle, d_sc, nterms, steps):
        """exactly `steps` MSMs, submitted back to back, each collected (result on the host, after the exchange for N>1) before returning.
        d_sc: one scalar buffer, or a pair — step i then takes buffer i % 2 and the last result of each is kept in last_of_set"""
        kms = []
        pair = isinstance(d_sc, (tuple, list))
        for i in range(steps + DEPTH):
            if i >= DEPTH:
                j = i - DEPTH; slot = j % NSLOT
                zk.check(L.zkt_g1_msm_collect(handle, slot, outp, None) if world == 1 else L.zkt_g1_msm_sharded_collect(handle, slot, outp))
                kms.append(L.zkt_last_kernel_ms())
                if pair: last_of_set[j % 2] = out.copy()
            if i < steps:
                zk.check(L.zkt_g1_msm_submit(handle, vp(d_sc[i % 2] if pair else d_sc), nterms, sp, i % NSLOT))
        return kms

    def timed(handle, d_sc, nterms, steps):
        if world > 1: dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kms = run(handle, d_sc, nterms, steps)
        torch.cuda.synchronize()
        if world > 1: dist.barrier()
        return max_over_ranks(time.perf_counter() - t0), kms

    run(h, d_scalars, n, args.warmup)
    # single-MSM latency (blocking call, nothing else in flight)
    latency_ms = None
    if world == 1:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        zk.check(L.zkt_g1_msm_dev(h, vp(d_scalars), n, sp, outp, None))
        latency_ms = (time.perf_counter() - t0) * 1e3
    elapsed, kern_ms = timed(h, (d_scalars, d_scalars_b), n, args.steps)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * n * args.steps / elapsed
    k_ms = float(np.mean(kern_ms))
    k_ms_note = None
    if not k_ms > 0:           # below 2^19 terms the library replays an MSM's pipeline as one graph launch: no HIP events around its accumulate kernel
        k_ms, k_ms_note = ms_per_step, "no per-kernel events at this size (graph replay): the whole-step time stands in for the kernel time, so `achieved` is a lower bound"
    headline_point = last_of_set[0] if last_of_set[0] is not None else out.copy()
    headline_point_b = last_of_set[1]

    # strong scaling beside the weak headline: ONE 2^log2n-term MSM cut into `world` index ranges
    strong = None
    if world > 1:
        ns = n // world
        hs = ctypes.c_void_p()
        zk.check(L.zkt_g1_bases_from_device(vp(d_bases), ns, sp, ctypes.byref(hs)))      # this rank's range: the first n/world of its bases
        d_ss = d_scalars[:ns].contiguous()
        run(hs, d_ss, ns, 2)
        el, _ = timed(hs, d_ss, ns, args.steps)
        strong = {"metric": "G1 MSM scalar-muls/sec, one 2^%d-term MSM over %d GPUs" % (args.log2n, world), "value": ns * world * args.steps / el, "ms_per_msm": el / args.steps * 1e3,
                  "terms_per_gpu": ns, "scaling": "strong",
                  "note": "below 2^19 terms per GPU every stage is latency-bound (sort + reduce tail ~2.5 ms): the speed-up over one GPU is bounded by that floor, not by the exchange (168 B per rank)"}
        L.zkt_g1_bases_free(hs)

    # ---- BASELINE config 4: Groth16 prove + verify at 2^20 constraints on the synthetic chain R1CS, sparse-R1CS path (row f-3).
    # N = 1: the whole proof on one GPU, two proofs in flight.  N > 1: (a) ONE proof sharded — every rank keeps an index range of the three resident
    # base sets and evaluates only its own range of the quotient (no exchange in the Fr stage), one all-gather of 672 B per proof inside zkt_groth16_prove_r1cs_sharded; (b) replicas — every rank
    # its own key and its own proofs, no exchange at all.
    g16 = None
    if args.groth16_log2n > 0:
        stage = "import"
        try:
            from qap_util import chain_circuit_sparse, sparse_struct, alloc_crs
            from zkt_testlib import SplitMix64, ints_to_arr, ptr
            gn = 1 << args.groth16_log2n
            mats, wires, gl, gm = chain_circuit_sparse(gn, seed=7)
            rng = SplitMix64(7)
            trap = [ints_to_arr([rng.below(R_MOD - 1) + 1], 4) for _ in range(5)]
            pr, ps = ints_to_arr([rng.below(R_MOD - 1) + 1], 4), ints_to_arr([rng.below(R_MOD - 1) + 1], 4)
            structs = [sparse_struct(*M) for M in mats]
            gp = (np.zeros((1, 13), np.uint64), np.zeros((1, 25), np.uint64), np.zeros((1, 13), np.uint64))
            outs = [x.ctypes.data for x in gp]
            d_w = torch.from_numpy(wires.view(np.int64)).to(dev)

            def setup(shard, nshards):
                vk, vbuf = alloc_crs(1, gl, gm); vk.g1_uvw_wit = None
                pk = ctypes.c_void_p()
                t0 = time.perf_counter()
                rc = L.zkt_groth16_setup_r1cs_sharded(gn, gl, gm, *[ctypes.addressof(x) for x in structs], *[t.ctypes.data for t in trap], shard, nshards,
                                                      ctypes.addressof(vk), ctypes.addressof(pk))
                return rc, pk, vk, vbuf, time.perf_counter() - t0

            def prove_pipelined(pk, k):
                """two proofs in flight on the key — the Fr stage of proof i+1 runs under the MSMs of proof i; every proof is collected on the host"""
                zk.check(L.zkt_groth16_prove_r1cs_submit(pk, 0, d_w.data_ptr(), pr.ctypes.data, ps.ctypes.data))
                for i in range(k):
                    if i + 1 < k: zk.check(L.zkt_groth16_prove_r1cs_submit(pk, (i + 1) % 2, d_w.data_ptr(), pr.ctypes.data, ps.ctypes.data))
                    zk.check(L.zkt_groth16_prove_r1cs_collect(pk, i % 2, *outs))

            def time_leg(fn):
                fn(1); torch.cuda.synchronize()
                if world > 1: dist.barrier()
                t0 = time.perf_counter()
                fn(args.groth16_proofs)
                torch.cuda.synchronize()
                if world > 1: dist.barrier()
                return max_over_ranks(time.perf_counter() - t0) / args.groth16_proofs

            g16 = {"metric": "Groth16 proofs/sec", "constraints": gn, "wires": gm + 1, "n_gpus": world, "proofs_timed": args.groth16_proofs,
                   "workload": "chain R1CS w_{j+1} = w_j^2 + c_j, witness resident in HBM, trapdoors and r,s injected"}
            vk = None
            if world > 1:
                stage = "sharded setup"
                rc, pk, vk, vbuf, g_setup = setup(rank, world)
                if not all_ok(rc == 0): raise RuntimeError("sharded setup failed on some rank (rc %d here)" % rc)
                stage = "sharded prove"

                def prove_sharded(k):
                    for _ in range(k): zk.check(L.zkt_groth16_prove_r1cs_sharded(pk, d_w.data_ptr(), pr.ctypes.data, ps.ctypes.data, *outs))
                dt = time_leg(prove_sharded)
                L.zkt_groth16_pk_free(pk)
                g16["sharded"] = {"value": 1.0 / dt, "ms_per_proof": dt * 1e3, "setup_s": round(g_setup, 2),
                                  "sharding": "index ranges of the three resident MSM base sets per rank; one all-gather of 672-B Jacobian partials per proof (C ABI, RCCL)",
                                  "amdahl_note": "every rank evaluates only its own range of the quotient (N forward transforms of 2n/N and one inverse per polynomial, no exchange: about half of a "
                                                 "replicated Fr stage); the MSM shards of 2^20/N terms sit on the latency floor of the sort + reduce stages and run more windows per term than one "
                                                 "2^20-term MSM, which is what bounds the speed-up (DESIGN.md §6)"}
                sharded_proof = [x.copy() for x in gp]
            stage = "setup"
            rc, pk, vk1, vbuf1, g_setup = setup(0, 1)
            if not all_ok(rc == 0): raise RuntimeError("setup failed on some rank (rc %d here)" % rc)
            stage = "prove"
            dt = time_leg(lambda k: prove_pipelined(pk, k))
            L.zkt_groth16_pk_free(pk)
            g16.update({"value": world / dt, "ms_per_proof": dt * 1e3, "proofs_in_flight": 2 * world, "setup_s": round(g_setup, 2),
                        "mode": "one key, two proofs in flight" if world == 1 else "replicas: one key and two proofs in flight per GPU, no exchange"})
            if world > 1:
                g16["sharded"]["equals_unsharded_proof"] = bool(all((a == b).all() for a, b in zip(sharded_proof, gp)))
            elif gn >= 64:
                # BASELINE config 4 on ONE card: what rank 3 of 8 does for one proof sharded over 8 GPUs (its index ranges of the three base sets, its own range of the quotient,
                # the three Jacobian partials; the exchange is 672 B more) — one proof at a time, against one unpipelined proof on this GPU.  Not a scaling measurement:
                # the per-rank time a node of 8 would run at (the eight ranks' partials combine to the unsharded proof: tests/test_gpu_fullsize.py).
                stage = "share of 8"
                rc, spk, svk, svbuf, s_setup = setup(3, 8)
                if rc != 0: raise RuntimeError("sharded setup (rank 3 of 8) failed: rc %d" % rc)
                d_parts = torch.zeros(1024, dtype=torch.int32, device=dev)
                for _ in range(2): zk.check(L.zkt_groth16_prove_r1cs_partials(spk, d_w.data_ptr(), pr.ctypes.data, ps.ctypes.data, d_parts.data_ptr()))
                torch.cuda.synchronize(); tsh = []
                for _ in range(7):
                    t0 = time.perf_counter(); zk.check(L.zkt_groth16_prove_r1cs_partials(spk, d_w.data_ptr(), pr.ctypes.data, ps.ctypes.data, d_parts.data_ptr())); tsh.append(time.perf_counter() - t0)
                L.zkt_groth16_pk_free(spk)
                rc, pk, vkx, vbufx, _ = setup(0, 1)
                if rc != 0: raise RuntimeError("setup failed: rc %d" % rc)
                for _ in range(2): zk.check(L.zkt_groth16_prove_r1cs_dev(pk, d_w.data_ptr(), pr.ctypes.data, ps.ctypes.data, *outs))
                tone = []
                for _ in range(5):
                    t0 = time.perf_counter(); zk.check(L.zkt_groth16_prove_r1cs_dev(pk, d_w.data_ptr(), pr.ctypes.data, ps.ctypes.data, *outs)); tone.append(time.perf_counter() - t0)
                L.zkt_groth16_pk_free(pk)
                g16["share_of_8"] = {"rank": 3, "ms_per_rank_median": sorted(tsh)[3] * 1e3, "ms_per_rank_best": min(tsh) * 1e3, "one_proof_unpipelined_ms": sorted(tone)[2] * 1e3,
                                     "ratio": sorted(tone)[2] / sorted(tsh)[3], "setup_s": round(s_setup, 2),
                                     "note": "one rank's share of ONE proof sharded over 8 GPUs, measured on this one card (its MSM shards + its own range of the quotient, no exchange in the Fr "
                                             "stage; the all-gather of 672 B is not included); ratio = one unpipelined proof on one GPU / this share — the latency ratio a node of 8 would reach, "
                                             "not a measured scaling curve"}
            if rank == 0:
                stmt = wires[:gl + 1].copy()
                # an UNPREPARED key: the library builds the key's entry for the 63-step loop at first sight, beside the first call (served by kernels that need nothing of
                # the key); every later call is a steady-state call.  zkt_groth16_vk_prepare (a caller that knows its key; ~5 ms) is timed by tools/bench_verify_latency.py.
                rc_prep = 0
                t0 = time.perf_counter()
                ok = L.zkt_groth16_verify(ctypes.byref(vk1), ptr(gp[0]), ptr(gp[1]), ptr(gp[2]), ptr(stmt), gl + 1)
                g16["verify_first_call_ms"] = (time.perf_counter() - t0) * 1e3; g16["verifies"] = bool(ok == 1) and rc_prep == 0
                t0 = time.perf_counter()
                ok2 = L.zkt_groth16_verify(ctypes.byref(vk1), ptr(gp[0]), ptr(gp[1]), ptr(gp[2]), ptr(stmt), gl + 1)
                g16["verify_second_call_ms"] = (time.perf_counter() - t0) * 1e3
                tv = []
                for _ in range(5):                       # steady state: every later verification against the key
                    t0 = time.perf_counter(); ok2 &= L.zkt_groth16_verify(ctypes.byref(vk1), ptr(gp[0]), ptr(gp[1]), ptr(gp[2]), ptr(stmt), gl + 1); tv.append(time.perf_counter() - t0)
                g16["verify_ms"] = sorted(tv)[2] * 1e3; g16["verifies"] = g16["verifies"] and bool(ok2 == 1)
        except Exception as e:      # never lose the headline line to the secondary leg; every collective above is preceded by an all-ranks agreement
            g16 = {"error": repr(e), "stage": stage}

    # HBM traffic and instruction counts of the dominant kernel come from separate rocprofv3 --pmc passes (committed summaries), never from this run
    traffic_prof = load_profile(PROFILE_ROUND + "_hbm_traffic_pmc.json") or load_profile("r02_hbm_traffic_pmc.json") or load_profile("r01_hbm_traffic_pmc.json")
    traffic = None
    if traffic_prof and args.log2n == 20:
        traffic = traffic_prof.get("k_accumulate_hbm_bytes_per_launch", {}).get("uncorrected")
    sq = load_profile(PROFILE_ROUND + "_accumulate_sq_counters.json") or load_profile("r02_accumulate_sq_counters.json")
    # a counter file is only as good as the kernels it was taken on: it carries the hash of the kernel sources (tools/src_hash.py); say so if they changed since
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from src_hash import kernel_sources_sha16
        here = kernel_sources_sha16()
    except Exception:
        here = None
    stale = lambda prof: None if not prof else (prof.get("kernel_sources_sha16") != here)
    valu = None
    if sq and args.log2n == 20 and "k_accumulate_g1" in sq:
        ka = sq["k_accumulate_g1"]
        lane_instr = ka["valu_lane_instr_per_launch"]                      # SQ_INSTS_VALU x 64 lanes, per launch
        peak = sq["valu_peak"]["int_mad_lane_ops_per_s_T"]                 # measured issue rate of v_mad_u64_u32 with >= 2 waves/SIMD (tools/ubench), not a datasheet figure
        vp_ = sq["valu_peak"]
        ach = lane_instr / (k_ms * 1e-3) / 1e12
        valu = {"achieved": ach, "peak": peak, "unit": "T lane-instr/s", "frac": ach / peak,
                "peak_note": "peak = measured chip-wide issue rate of v_mad_u64_u32 (two thirds of this kernel's instructions); the same harness sees v_fma_f32 at %.1f T/s, "
                             "so the integer multiply-add issues at under half the f32 rate" % vp_["f32_fma_lane_ops_per_s_T"],
                "frac_of_f32_fma_rate": ach / vp_["f32_fma_lane_ops_per_s_T"],
                "field_mix_rate_at_this_occupancy_T": vp_.get("field_mix_at_2_waves_per_simd_T"),
                "lane_instr_per_bucket_add": lane_instr / (13 * n),
                "wait_any_frac": ka.get("wait_any_frac_of_wave_cycles"), "counters_stale": stale(sq),
                "source": "profiles/%s_accumulate_sq_counters.json (rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES ..., folded by tools/sq_summary.py), peaks from profiles/%s_valu_ubench.txt" % (PROFILE_ROUND, PROFILE_ROUND)}
    result = {
        "metric": "G1 MSM scalar-muls/sec at 2^%d bases per GPU (BLS12-381)" % args.log2n,
        "value": value, "unit": "scalar-muls/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32 limbs (384-bit Montgomery integer)", "data": "synthetic",
        "config": {"workload": "Batched G1 Pippenger MSM, 2^%d random bases/scalars per GPU, bases device-resident with window multiples" % args.log2n,
                   "terms_per_gpu": n, "scalar_bits": 255, "scalar_dist": args.scalar_dist,
                   "sharding": "index range per rank; all-gather of 168-B Jacobian partials + on-device combine inside the C ABI (zkt_g1_msm_sharded_collect)" if world > 1 else "none",
                   "exchange_transport": transport,
                   "bases_setup_s": round(setup_s, 3), "msms_in_flight": DEPTH,
                   "single_msm_latency_ms": round(latency_ms, 3) if latency_ms is not None else None},
        "roofline": {"bound": "hbm", "kernel": L.zkt_last_kernel_name().decode(),
                     "achieved": MSM_BYTES_PER_TERM * n / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": MSM_BYTES_PER_TERM * n / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_counters_stale": stale(traffic_prof),
                     "traffic_note": "FETCH_SIZE+WRITE_SIZE per launch from the committed rocprofv3 --pmc summary; a multiple of the algorithmic bytes by design: every term is gathered once per window from the resident window-multiple table (DESIGN.md §4)",
                     "kernel_ms": k_ms, **({"kernel_ms_note": k_ms_note} if k_ms_note else {}), "algorithmic_bytes_per_launch": MSM_BYTES_PER_TERM * n,
                     "note": "integer-VALU bound by construction (SURVEY §8d); `valu` is the roof that binds, from hardware counters",
                     "valu": valu},
    }
    if strong is not None: result["strong"] = strong

    # parity check of the timed configuration at full size, by linearity: bases are k_i*G, so the MSM over all ranks
    # must equal (sum_i k_i s_i mod r)*G — python integers only, independent of the HIP path
    k_ints = [int.from_bytes(a.tobytes(), "little") for a in h_k]
    tot = sum(k * int.from_bytes(b.tobytes(), "little") for k, b in zip(k_ints, h_scalars)) % R_MOD
    tot_b = sum(k * int.from_bytes(b.tobytes(), "little") for k, b in zip(k_ints, h_scalars_b)) % R_MOD
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, (tot, tot_b))
        tot = sum(p[0] for p in parts) % R_MOD; tot_b = sum(p[1] for p in parts) % R_MOD
    # N > 1: the pairing metric over all GPUs — independent batches per GPU, index-range partition, no exchange; rank 0's own leg below keeps the details
    pairing_all = None
    if world > 1 and args.pairings > 0:
        mq = min(args.pairings, n); okp = True; dtp = 0.0
        try:
            from zkt_testlib import G2_GEN
            g2a = np.zeros((1, 25), dtype=np.uint64)
            (x1, x0), (y1, y0) = G2_GEN
            g2a[0, 0:6] = int_to_limbs(x1, 6); g2a[0, 6:12] = int_to_limbs(x0, 6); g2a[0, 12:18] = int_to_limbs(y1, 6); g2a[0, 18:24] = int_to_limbs(y0, 6)
            a_g2 = torch.from_numpy(np.repeat(g2a, mq, axis=0).view(np.int64)).to(dev)
            a_kq = torch.from_numpy(rand_scalars_mod_r(600 + rank, mq).view(np.int64)).to(dev)
            a_q = torch.empty((mq, 25), dtype=torch.int64, device=dev)
            zk.check(L.zkt_g2_mul_batch_dev(vp(a_g2), vp(a_kq), 4, vp(a_q), mq, sp))
            a_p = d_bases[:mq].contiguous(); a_e = torch.empty((mq, 72), dtype=torch.int64, device=dev)
            zk.check(L.zkt_tate_batch_dev(vp(a_p), vp(a_q), vp(a_e), mq, sp)); torch.cuda.synchronize()          # warm
        except Exception as ex:
            okp = False; print("rank %d: pairing leg setup failed: %r" % (rank, ex), file=sys.stderr)
        if all_ok(okp):
            dist.barrier(); t0 = time.perf_counter()
            try:
                zk.check(L.zkt_tate_batch_dev(vp(a_p), vp(a_q), vp(a_e), mq, sp)); torch.cuda.synchronize()
            except Exception as ex:
                okp = False; print("rank %d: pairing leg failed: %r" % (rank, ex), file=sys.stderr)
            dtp = max_over_ranks(time.perf_counter() - t0)
            if all_ok(okp):
                pairing_all = {"metric": "Tate pairings/sec over all GPUs: an independent batch per GPU, no exchange", "value": world * mq / dtp, "n_gpus": world,
                               "batch_per_gpu": mq, "scaling": "weak", "ms": dtp * 1e3}
    failed = False
    if rank == 0:
        if pairing_all is not None: result["pairing_all_gpus"] = pairing_all
        want = g1_arr([py_g1_mul(G1_GEN, tot)])          # plain python-integer affine arithmetic: independent of the HIP path and of oracle/
        ok_a = (want == headline_point).all()
        ok_b = headline_point_b is None or (g1_arr([py_g1_mul(G1_GEN, tot_b)]) == headline_point_b).all()
        result["config"]["full_size_check"] = "ok" if (ok_a and ok_b) else "MISMATCH"
        result["config"]["full_size_check_note"] = "the timed loop alternates two scalar sets; the last result of each is compared with (sum k_i s_i mod r) G computed in python integers"
        failed = failed or result["config"]["full_size_check"] != "ok"

        # the one-shot host-pointer call at the same size (no resident table: PCIe + table-free plan), so the cost of residency is visible
        if world == 1:
            hb = d_bases.cpu().numpy().view(np.uint64)
            o1 = np.zeros((1, 13), dtype=np.uint64)
            zk.check(L.zkt_g1_msm(hb.ctypes.data_as(ctypes.c_void_p), h_scalars.ctypes.data_as(ctypes.c_void_p), n, o1.ctypes.data_as(ctypes.c_void_p)))
            t0 = time.perf_counter()
            zk.check(L.zkt_g1_msm(hb.ctypes.data_as(ctypes.c_void_p), h_scalars.ctypes.data_as(ctypes.c_void_p), n, o1.ctypes.data_as(ctypes.c_void_p)))
            result["config"]["one_shot_msm_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
            result["config"]["one_shot_note"] = "zkt_g1_msm with host pointers: upload of 2^20 x 136 B over PCIe + the table-free plan; the resident form pays bases_setup_s once instead"
            failed = failed or not (o1 == headline_point).all()

        # secondary metric: Tate pairings/s
        if args.pairings > 0:
            m = args.pairings
            from zkt_testlib import G2_GEN
            g2 = np.zeros((1, 25), dtype=np.uint64)
            (x1, x0), (y1, y0) = G2_GEN
            g2[0, 0:6] = int_to_limbs(x1, 6); g2[0, 6:12] = int_to_limbs(x0, 6); g2[0, 12:18] = int_to_limbs(y1, 6); g2[0, 18:24] = int_to_limbs(y0, 6)
            d_g2 = torch.from_numpy(np.repeat(g2, m, axis=0).view(np.int64)).to(dev)
            d_kq = torch.from_numpy(rand_scalars_mod_r(6, m).view(np.int64)).to(dev)
            d_q = torch.empty((m, 25), dtype=torch.int64, device=dev)
            zk.check(L.zkt_g2_mul_batch_dev(vp(d_g2), vp(d_kq), 4, vp(d_q), m, sp))
            d_p = d_bases[:m].contiguous()
            d_e = torch.empty((m, 72), dtype=torch.int64, device=dev)
            zk.check(L.zkt_tate_batch_dev(vp(d_p), vp(d_q), vp(d_e), m, sp))   # warm
            tps, kps = [], []
            for _ in range(5):                         # five timed launches, the median is the figure
                torch.cuda.synchronize(); t0 = time.perf_counter()
                zk.check(L.zkt_tate_batch_dev(vp(d_p), vp(d_q), vp(d_e), m, sp))
                torch.cuda.synchronize(); tps.append(time.perf_counter() - t0); kps.append(L.zkt_last_kernel_ms())
            dt = sorted(tps)[2]; k_tate_ms = sorted(kps)[2]
            # latency of ONE pairing: batches this small take the lane-distributed kernel (csrc/zkt_dpairing.hip, one pairing per 12 lanes)
            zk.check(L.zkt_tate_batch_dev(vp(d_p), vp(d_q), vp(d_e), 1, sp)); torch.cuda.synchronize()
            t0 = time.perf_counter(); zk.check(L.zkt_tate_batch_dev(vp(d_p), vp(d_q), vp(d_e), 1, sp)); torch.cuda.synchronize(); lat1 = time.perf_counter() - t0
            zk.check(L.zkt_tate_batch_dev(vp(d_p), vp(d_q), vp(d_e), m, sp)); torch.cuda.synchronize()          # restore the full batch for the CPU comparison below
            pt = load_profile(PROFILE_ROUND + "_tate_sq_counters.json") or load_profile("r02_tate_sq_counters.json") or load_profile("r01_tate_sq_counters.json") or {}
            pinstr = pt.get("valu_instr_per_pairing") or pt.get("valu_instructions_per_pairing") or (pt.get("kernels", {}).get("k_tate", {}).get("valu_instr_per_wave"))
            tmem = load_profile(PROFILE_ROUND + "_tate_memory_counters.json") or load_profile("r03_tate_memory_counters.json") or {}
            result["pairing"] = {"metric": "Tate pairings/sec", "value": m / dt, "batch": m, "kernel_ms": k_tate_ms, "launches_timed": 5, "ms_all": [round(t * 1e3, 3) for t in tps],
                                 "roofline": {"bound": "hbm", "kernel": "k_tate", "achieved": PAIRING_BYTES * m / (k_tate_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": PAIRING_BYTES * m / (k_tate_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": PAIRING_BYTES * m,
                                              "traffic": (lambda kt: (kt.get("FETCH_SIZE_KB_avg_per_launch", 0) + kt.get("WRITE_SIZE_KB_avg_per_launch", 0)) * 1024 if kt else None)((tmem.get("kernels", {}) or {}).get("k_tate")) if m == 1 << 16 else None, "traffic_counters_stale": stale(tmem) if tmem else None,
                                              "note": "integer-VALU bound (pairing.valu); the traffic is the per-lane scratch frame of the Fq12 temporaries, not algorithmic bytes"},
                                 "hbm_achieved_GBs": PAIRING_BYTES * m / dt / 1e9, "hbm_frac": PAIRING_BYTES * m / dt / 1e9 / HBM_PEAK_GBS,
                                 "single_pairing_latency_ms": round(lat1 * 1e3, 3),
                                 "small_batch_note": "n <= 24,576 runs one pairing per 12 lanes (lane-distributed kernel): ~5 ms for 1..2048 pairings; larger batches one pairing per lane"}
            if pinstr and sq:
                peak = sq["valu_peak"]["int_mad_lane_ops_per_s_T"]
                result["pairing"]["valu"] = {"achieved": pinstr * m / dt / 1e12, "peak": peak, "unit": "T lane-instr/s", "frac": pinstr * m / dt / 1e12 / peak,
                                             "source": "SQ_INSTS_VALU / SQ_WAVES of the pairing kernel (committed rocprofv3 --pmc summary)"}

        # The B sum of a Groth16 proof (eval_with_g2_hidings, polynomial.rs:283-293): a resident G2 MSM at the proof's size — the largest slice of a proof.
        # Pipelined like the headline; checked at full size by linearity in python integers; VALU / traffic from the committed counter passes.
        if world == 1 and args.g2_log2n > 0:
            try:
                from zkt_testlib import G2_GEN, py_g2_mul, g2_arr, to_abi_g2
                gn2 = 1 << args.g2_log2n
                g2g = np.zeros((1, 25), dtype=np.uint64)
                (x1, x0), (y1, y0) = G2_GEN
                g2g[0, 0:6] = int_to_limbs(x1, 6); g2g[0, 6:12] = int_to_limbs(x0, 6); g2g[0, 12:18] = int_to_limbs(y1, 6); g2g[0, 18:24] = int_to_limbs(y0, 6)
                hk2 = rand_scalars_mod_r(31, gn2); hs2 = rand_scalars_mod_r(32, gn2)
                t_gen = torch.from_numpy(np.repeat(g2g, gn2, axis=0).view(np.int64)).to(dev); t_k = torch.from_numpy(hk2.view(np.int64)).to(dev)
                t_b = torch.empty((gn2, 25), dtype=torch.int64, device=dev)
                zk.check(L.zkt_g2_mul_batch_dev(vp(t_gen), vp(t_k), 4, vp(t_b), gn2, sp)); torch.cuda.synchronize(); del t_gen
                h2 = ctypes.c_void_p(); t0 = time.perf_counter(); zk.check(L.zkt_g2_bases_from_device(vp(t_b), gn2, sp, ctypes.byref(h2))); t_setup2 = time.perf_counter() - t0
                t_s = torch.from_numpy(hs2.view(np.int64)).to(dev)
                o2 = np.zeros((1, 25), np.uint64); o2p = o2.ctypes.data_as(ctypes.c_void_p)

                def run2(k, depth=3):
                    for i in range(k + depth):
                        if i >= depth: zk.check(L.zkt_g2_msm_collect(h2, (i - depth) % NSLOT, o2p, None))
                        if i < k: zk.check(L.zkt_g2_msm_submit(h2, vp(t_s), gn2, sp, i % NSLOT))
                run2(2); torch.cuda.synchronize()
                t0 = time.perf_counter(); run2(8); torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / 8
                zk.check(L.zkt_g2_msm_dev(h2, vp(t_s), gn2, sp, o2p, None)); torch.cuda.synchronize()
                t0 = time.perf_counter(); zk.check(L.zkt_g2_msm_dev(h2, vp(t_s), gn2, sp, o2p, None)); lat2 = time.perf_counter() - t0
                tot2 = sum(int.from_bytes(a.tobytes(), "little") * int.from_bytes(b.tobytes(), "little") for a, b in zip(hk2, hs2)) % R_MOD
                ok2 = bool((o2 == g2_arr([to_abi_g2(py_g2_mul(((x0, x1), (y0, y1)), tot2))])).all())
                L.zkt_g2_bases_free(h2); del t_b, t_s, t_k
                g2sq = load_profile(PROFILE_ROUND + "_g2_msm_sq_counters.json") or load_profile("r02_g2_msm_sq_counters.json") or {}
                g2mem = load_profile(PROFILE_ROUND + "_g2_msm_memory_counters.json") or {}
                g2k = next((v for k, v in (g2sq.get("kernels") or {}).items() if "k_accumulate_g2_pair" in k and "direct" not in k), None)
                g2m = next((v for k, v in (g2mem.get("kernels") or {}).items() if "k_accumulate_g2_pair" in k and "direct" not in k), None)
                g2valu = None
                if g2k and sq and args.g2_log2n == 20 and g2k.get("valu_lane_instr_per_launch"):
                    pk = sq["valu_peak"]["int_mad_lane_ops_per_s_T"]
                    g2valu = {"kernel": "k_accumulate_g2_pair", "lane_instr_per_launch": g2k["valu_lane_instr_per_launch"], "lane_instr_per_bucket_add": g2k["valu_lane_instr_per_launch"] / (13 * gn2),
                              "vgpr": g2k.get("vgpr"), "wait_any_frac": g2k.get("wait_any_frac_of_wave_cycles"),
                              "achieved_lower_bound": g2k["valu_lane_instr_per_launch"] / (dt2) / 1e12, "peak": pk, "unit": "T lane-instr/s", "frac_lower_bound": g2k["valu_lane_instr_per_launch"] / dt2 / 1e12 / pk,
                              "note": "counted lane-instructions of ONE accumulate launch over the whole pipelined step time (the kernel's own time is shorter: a lower bound of its issue rate)",
                              "traffic_bytes_per_launch": ((g2m.get("FETCH_SIZE_KB_avg_per_launch", 0) + g2m.get("WRITE_SIZE_KB_avg_per_launch", 0)) * 1024) if g2m else None}
                g2blk = {"metric": "G2 MSM scalar-muls/sec at 2^%d bases (resident, pipelined)" % args.g2_log2n, "value": gn2 / dt2, "ms_per_msm": dt2 * 1e3, "single_msm_latency_ms": round(lat2 * 1e3, 3),
                         "bases_setup_s": round(t_setup2, 3), "full_size_check": "ok" if ok2 else "MISMATCH",
                         "plan": "sort -> XYZZ accumulate on lane pairs (k_accumulate_g2_pair) -> four-lane reduce; affine pair-tree rounds: %s (ZKT_G2_AFFINE_ROUNDS, off by default: profiles/r04_batched_affine_go_no_go.md)" % os.environ.get("ZKT_G2_AFFINE_ROUNDS", "0"),
                         "roofline": {"bound": "hbm", "achieved": 224 * gn2 / dt2 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 224 * gn2 / dt2 / 1e9 / HBM_PEAK_GBS,
                                      "algorithmic_bytes_per_msm": 224 * gn2, "note": "192 B base + 32 B scalar per term over the whole pipelined step (sort, accumulate and reduce kernels of consecutive MSMs overlap); integer-VALU bound",
                                      "valu": g2valu, "counters_stale": stale(g2sq) if g2sq else None}}
                result["g2_msm"] = g2blk
                failed = failed or not ok2
            except Exception as e:
                result["g2_msm"] = {"error": repr(e)}

        # BASELINE config 5: Bulletproofs range proof over 65,536 bits (64 bits x 1024 values) and its inner-product argument, generators resident
        if world == 1 and args.bulletproofs:
            try:
                from zkt_testlib import SECP_N, rand_u64_array, ints_to_arr, ptr as _p, SplitMix64 as _SM
                bn = 1 << 16
                g0 = np.zeros((1, 9), np.uint64); L.zkt_secp_generator(g0.ctypes.data_as(ctypes.c_void_p))
                ks = rand_u64_array(11, (2 * bn + 3, 4)); ks[:, 3] >>= np.uint64(1)
                pts = np.zeros((2 * bn + 3, 9), np.uint64)
                zk.check(L.zkt_secp_mul_batch(_p(np.repeat(g0, 2 * bn + 3, axis=0)), _p(ks), 4, _p(pts), 2 * bn + 3))
                gg, hh, uu, g_r, h_r = pts[:bn].copy(), pts[bn:2 * bn].copy(), pts[2 * bn:2 * bn + 1].copy(), pts[2 * bn + 1:2 * bn + 2].copy(), pts[2 * bn + 2:].copy()
                bits = [int(v) for v in (rand_u64_array(15, (bn,)) & np.uint64(1))]
                val = sum(bt << i for i, bt in enumerate(bits)); aL = ints_to_arr(bits, 4); gam = ints_to_arr([_SM(17).below(SECP_N)], 4)
                tmp2, Vv = np.zeros((2, 9), np.uint64), np.zeros((1, 9), np.uint64)
                zk.check(L.zkt_secp_mul_batch(_p(np.concatenate([g_r, h_r])), _p(np.concatenate([ints_to_arr([val % SECP_N], 4), gam])), 4, _p(tmp2), 2))
                zk.check(L.zkt_secp_add_batch(_p(tmp2[0:1].copy()), _p(tmp2[1:2].copy()), _p(Vv), 1))
                rnd = rand_u64_array(18, (7 + 2 * bn, 4)); rnd[:, 3] >>= np.uint64(1); rnd[:, 0] |= np.uint64(1)
                xs = rand_u64_array(14, (16, 4)); xs[:, 3] >>= np.uint64(1); xs[:, 0] |= np.uint64(1)
                bctx = ctypes.c_void_p(); t0 = time.perf_counter(); zk.check(L.zkt_bp_ipa_ctx_create(bn, _p(gg), _p(hh), _p(uu), ctypes.byref(bctx))); t_ctx = time.perf_counter() - t0
                bp = {"metric": "Bulletproofs range proofs/sec, 65,536 bits (64-bit x 1024 values), generators resident", "generator_setup_ms": round(t_ctx * 1e3, 2)}
                for use_ipa in (0, 1):
                    okp = L.zkt_bp_range_proof_ctx(bctx, _p(Vv), _p(aL), _p(gam), _p(g_r), _p(h_r), use_ipa, _p(rnd), _p(xs), None)
                    ts = []
                    for _ in range(8):         # the first call on a context captures its MSM pipelines as graphs; the best of the following replays is the figure
                        t0 = time.perf_counter(); okp &= L.zkt_bp_range_proof_ctx(bctx, _p(Vv), _p(aL), _p(gam), _p(g_r), _p(h_r), use_ipa, _p(rnd), _p(xs), None); ts.append(time.perf_counter() - t0)
                    bp["ms_with_ipa" if use_ipa else "ms_without_ipa"] = round(min(ts) * 1e3, 3)
                    bp["accepts"] = bool(okp == 1) and bp.get("accepts", True)
                bad = aL.copy(); bad[777, 0] ^= np.uint64(1)
                bp["rejects_wrong_opening"] = bool(L.zkt_bp_range_proof_ctx(bctx, _p(Vv), _p(bad), _p(gam), _p(g_r), _p(h_r), 1, _p(rnd), _p(xs), None) == 0)
                bp["value"] = 1e3 / bp["ms_with_ipa"]
                L.zkt_bp_ipa_ctx_free(bctx)
                result["bulletproofs"] = bp
                failed = failed or not (bp["accepts"] and bp["rejects_wrong_opening"])
            except Exception as e:
                result["bulletproofs"] = {"error": repr(e)}
        if g16 is not None:
            result["groth16"] = g16
            failed = failed or g16.get("verifies") is False or "error" in g16

        # CPU baseline: the oracle (faithful restatement of the reference algorithm) on a bounded sample
        if world == 1 and not args.no_cpu:
            from zkt_testlib import oracle, ptr
            O = oracle()
            cores = min(os.cpu_count() or 1, 16)      # the GPU box's CPU share for one GPU is 16 cores

            def cpu_msm(m, threads):
                pts = d_bases[:m].cpu().numpy().view(np.uint64).copy(); sc = h_scalars[:m].copy()
                tmp = np.zeros_like(pts); acc = np.zeros((1, 13), dtype=np.uint64); acc[0, 12] = 1
                t0 = time.perf_counter()
                assert O.zkto_g1_mul_batch(ptr(pts), ptr(sc), 4, ptr(tmp), m, threads) == 0      # n scalar-muls (macros.rs:1-32)
                for i in range(m):                                                               # n sequential affine adds (polynomial.rs:277-279)
                    O.zkto_g1_add_batch(ptr(acc), ptr(tmp[i:i + 1]), ptr(acc), 1)
                return time.perf_counter() - t0
            m = 2048 * cores                           # ~10 s of CPU work on 16 threads
            dt = cpu_msm(m, cores)
            m1 = 768                                   # the reference itself is single-threaded: ~4 s on one thread
            dt1 = cpu_msm(m1, 1)
            result["cpu_baseline"] = {"value": m / dt, "unit": "scalar-muls/s", "cores": cores, "kind": "port",
                                      "sample": "%d-term MSM by the oracle's reference algorithm (per-term double-and-add on %d threads + sequential affine adds), %.1f s" % (m, cores, dt),
                                      "single_thread": {"value": m1 / dt1, "unit": "scalar-muls/s", "cores": 1,
                                                        "sample": "%d-term MSM, same algorithm on one thread (the reference is single-threaded), %.1f s" % (m1, dt1)}}
            if "pairing" in result:                    # the reference's tate(): textbook Miller loop + 4314-bit exponentiation, eight pairings per thread
                mp = 8 * cores
                pp = d_bases[:mp].cpu().numpy().view(np.uint64).copy(); qq = d_q[:mp].cpu().numpy().view(np.uint64).copy()
                oo = np.zeros((mp, 72), dtype=np.uint64); idx = ctypes.c_size_t(0)
                t0 = time.perf_counter()
                assert O.zkto_pairing_batch(3, ptr(pp), ptr(qq), ptr(oo), mp, cores, ctypes.byref(idx)) == 0
                dtp = time.perf_counter() - t0
                t0 = time.perf_counter()
                assert O.zkto_pairing_batch(3, ptr(pp[:2].copy()), ptr(qq[:2].copy()), ptr(np.zeros((2, 72), dtype=np.uint64)), 2, 1, ctypes.byref(idx)) == 0
                dtp1 = time.perf_counter() - t0
                same = bool((oo == d_e[:mp].cpu().numpy().view(np.uint64)).all())
                result["pairing"]["cpu_baseline"] = {"value": mp / dtp, "unit": "pairings/s", "cores": cores, "kind": "port",
                                                     "sample": "%d Tate pairings by the oracle's reference algorithm on %d threads, %.1f s" % (mp, cores, dtp),
                                                     "single_thread": {"value": 2 / dtp1, "unit": "pairings/s", "cores": 1, "sample": "2 pairings on one thread, %.1f s" % dtp1},
                                                     "matches_gpu_bits": same}
                failed = failed or not same
        print(json.dumps(result), flush=True)
    L.zkt_g1_bases_free(h)
    if world > 1:
        L.zkt_comm_finalize()
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.exit(1)


if __name__ == "__main__":
    main()
