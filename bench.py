#!/usr/bin/env python3
"""bench.py — G1 MSM throughput at 2^20 bases per GPU (BASELINE.json configs[1]), plus the
Tate-pairing rate, on MI355X.  Contract: python bench.py --gpus N --steps K --warmup W
(N>1 under torch.distributed.run, one rank per GPU, RCCL).  One JSON line on rank 0.

step  = one MSM over 2^20 device-resident bases (a CRS) and 2^20 device-resident 255-bit
        scalars, result normalised to an affine point on the host.
N>1   = each rank owns a 2^20-term shard of one N*2^20-term MSM (weak scaling); the only
        exchange is an all_gather of the 168-byte Jacobian partial sums + a local add.
value = total scalar-muls per second over all ranks (terms / wall time, max over ranks)."""
import argparse, ctypes, importlib, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # the MSM pipeline wants its three stage streams on distinct hardware queues
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
PAIRING_VALU_INSTR = 23.13e6    # k_tate: VALU instructions per pairing (measured: SQ_INSTS_VALU / SQ_WAVES)
VALU_INSTR_PER_ADD, NWIN_2P20 = 5450, 13     # k_accumulate: instructions per XYZZ mixed add on the hot path (ISA count, 3.6 k of them v_mad_u64_u32); windows at n = 2^20 (c = 20)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MSM_BYTES_PER_TERM = 128       # 96 B affine base + 32 B scalar
PAIRING_BYTES = 864            # 96 + 192 in, 576 out


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
    ap.add_argument("--groth16-log2n", type=int, default=20, help="constraints (log2) of the Groth16 prove+verify leg at N=1 (0 = skip)")
    ap.add_argument("--groth16-proofs", type=int, default=8)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
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

    zk = importlib.import_module("zk-toolkit_amd")
    sharded = importlib.import_module("zk-toolkit_amd.sharded")
    zk.init(local)
    L = zk.lib()
    stream = torch.cuda.current_stream()
    sp = ctypes.c_void_p(stream.cuda_stream)
    n = 1 << args.log2n
    vp = lambda t: ctypes.c_void_p(t.data_ptr())

    # ---- synthetic inputs, resident in HBM: P_i = k_i * G1 (seed 3 + rank), s_i uniform in [0,r) (seed 4 + rank)
    from zkt_testlib import G1_GEN, int_to_limbs
    gen = np.zeros((1, 13), dtype=np.uint64); gen[0, :6] = int_to_limbs(G1_GEN[0], 6); gen[0, 6:12] = int_to_limbs(G1_GEN[1], 6)
    d_gen = torch.from_numpy(np.repeat(gen, n, axis=0).view(np.int64)).to(dev)
    d_k = torch.from_numpy(rand_scalars_mod_r(3 + 1000 * rank, n).view(np.int64)).to(dev)
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
    d_partial = torch.zeros(zk.G1_PARTIAL_WORDS, dtype=torch.int32, device=dev)
    out = np.zeros((1, 13), dtype=np.uint64)
    outp = out.ctypes.data_as(ctypes.c_void_p)

    DEPTH = int(os.environ.get("ZKT_BENCH_DEPTH", "5"))   # MSMs in flight: sort / accumulate / reduce-tail of consecutive MSMs overlap
    NSLOT = 8          # ZKT_MSM_SLOTS

    def finish(slot):
        """collect one MSM; for N>1 also the exchange step: all_gather of the 168-B partials + local add."""
        if world == 1:
            zk.check(L.zkt_g1_msm_collect(h, slot, outp, None))
        else:
            zk.check(L.zkt_g1_msm_collect(h, slot, None, vp(d_partial)))
            if backend == "nccl":
                g = sharded.sharded_sum(d_partial, lambda stack: stack.contiguous())      # RCCL all_gather of the 168-B partials
            else:
                g = sharded.sharded_sum(d_partial.cpu(), lambda stack: stack.contiguous()).to(dev)
            torch.cuda.current_stream().synchronize()
            zk.check(L.zkt_g1_jac_sum_dev(vp(g), world, sp, outp))
        return L.zkt_last_kernel_ms()

    def run(steps):
        """exactly `steps` MSMs, submitted back to back, each collected (result on the host) before returning"""
        kms = []
        for i in range(steps + DEPTH):
            if i >= DEPTH:
                kms.append(finish((i - DEPTH) % NSLOT))
            if i < steps:
                zk.check(L.zkt_g1_msm_submit(h, vp(d_scalars), n, sp, i % NSLOT))
        return kms

    run(args.warmup)
    # single-MSM latency (blocking call, nothing else in flight)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    zk.check(L.zkt_g1_msm_dev(h, vp(d_scalars), n, sp, outp, None) if world == 1 else 0)
    latency_ms = (time.perf_counter() - t0) * 1e3
    if world > 1: dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kern_ms = run(args.steps)
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * n * args.steps / elapsed
    k_ms = float(np.mean(kern_ms))

    # BASELINE config 4: Groth16 prove + verify at 2^20 constraints on the synthetic chain R1CS, sparse-R1CS path (row f-3).
    # N = 1: the whole proof on one GPU.  N > 1: every rank keeps an index range of the three resident base sets, the Fr stage is
    # replicated, and the only exchange is an all_gather of the three Jacobian partials (672 B) + a local combine — all ranks take part.
    g16 = None
    if args.groth16_log2n > 0:
        try:
            from qap_util import chain_circuit_sparse, sparse_struct, alloc_crs
            from zkt_testlib import SplitMix64, ints_to_arr, ptr
            gn = 1 << args.groth16_log2n
            mats, wires, gl, gm = chain_circuit_sparse(gn, seed=7)
            rng = SplitMix64(7)
            trap = [ints_to_arr([rng.below(R_MOD - 1) + 1], 4) for _ in range(5)]
            pr, ps = ints_to_arr([rng.below(R_MOD - 1) + 1], 4), ints_to_arr([rng.below(R_MOD - 1) + 1], 4)
            structs = [sparse_struct(*M) for M in mats]
            vk, vbuf = alloc_crs(1, gl, gm); vk.g1_uvw_wit = None
            pk = ctypes.c_void_p()
            t0 = time.perf_counter()
            zk.check(L.zkt_groth16_setup_r1cs_sharded(gn, gl, gm, *[ctypes.addressof(x) for x in structs], *[t.ctypes.data for t in trap], rank, world,
                                                       ctypes.addressof(vk), ctypes.addressof(pk)))
            g_setup = time.perf_counter() - t0
            gp = (np.zeros((1, 13), np.uint64), np.zeros((1, 25), np.uint64), np.zeros((1, 13), np.uint64))
            d_w = torch.from_numpy(wires.view(np.int64)).to(dev)
            d_part = torch.zeros(zk.GROTH16_PARTIAL_WORDS, dtype=torch.int32, device=dev)
            wa, wb = zk.G1_PARTIAL_WORDS, zk.G1_PARTIAL_WORDS + zk.G2_PARTIAL_WORDS

            def prove():
                if world == 1:
                    zk.check(L.zkt_groth16_prove_r1cs_dev(pk, d_w.data_ptr(), pr.ctypes.data, ps.ctypes.data, *[x.ctypes.data for x in gp]))
                    return
                zk.check(L.zkt_groth16_prove_r1cs_partials(pk, d_w.data_ptr(), pr.ctypes.data, ps.ctypes.data, d_part.data_ptr()))
                if backend == "nccl":
                    g = sharded.sharded_sum(d_part, lambda stack: stack.contiguous())
                else:
                    g = sharded.sharded_sum(d_part.cpu(), lambda stack: stack.contiguous()).to(dev)
                pa, pb, pc = g[:, :wa].contiguous(), g[:, wa:wb].contiguous(), g[:, wb:].contiguous()
                torch.cuda.current_stream().synchronize()
                zk.check(L.zkt_g1_jac_sum_dev(vp(pa), world, sp, ptr(gp[0])))
                zk.check(L.zkt_g2_jac_sum_dev(vp(pb), world, sp, ptr(gp[1])))
                zk.check(L.zkt_g1_jac_sum_dev(vp(pc), world, sp, ptr(gp[2])))

            def prove_pipelined(k):
                """N = 1: two proofs in flight on the key — the Fr stage of proof i+1 runs under the MSMs of proof i; every proof is collected on the host"""
                outs = [x.ctypes.data for x in gp]
                zk.check(L.zkt_groth16_prove_r1cs_submit(pk, 0, d_w.data_ptr(), pr.ctypes.data, ps.ctypes.data))
                for i in range(k):
                    if i + 1 < k: zk.check(L.zkt_groth16_prove_r1cs_submit(pk, (i + 1) % 2, d_w.data_ptr(), pr.ctypes.data, ps.ctypes.data))
                    zk.check(L.zkt_groth16_prove_r1cs_collect(pk, i % 2, *outs))

            prove(); torch.cuda.synchronize()
            if world > 1: dist.barrier()
            t0 = time.perf_counter()
            if world == 1: prove_pipelined(args.groth16_proofs)
            else:
                for _ in range(args.groth16_proofs): prove()
            torch.cuda.synchronize()
            if world > 1: dist.barrier()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            dt /= args.groth16_proofs
            L.zkt_groth16_pk_free(pk)
            g16 = {"metric": "Groth16 proofs/sec", "value": 1.0 / dt, "constraints": gn, "wires": gm + 1, "ms_per_proof": dt * 1e3, "n_gpus": world,
                   "proofs_in_flight": 2 if world == 1 else 1, "sharding": "none" if world == 1 else "index ranges of the three resident MSM base sets per rank; all_gather of 672-B Jacobian partials per proof",
                   "setup_s": round(g_setup, 2), "proofs_timed": args.groth16_proofs,
                   "workload": "chain R1CS w_{j+1} = w_j^2 + c_j, witness resident in HBM, trapdoors and r,s injected"}
            if rank == 0:
                stmt = wires[:gl + 1].copy()
                t0 = time.perf_counter()
                ok = L.zkt_groth16_verify(ctypes.byref(vk), ptr(gp[0]), ptr(gp[1]), ptr(gp[2]), ptr(stmt), gl + 1)
                g16["verify_ms"] = (time.perf_counter() - t0) * 1e3; g16["verifies"] = bool(ok == 1)
        except Exception as e:      # never lose the headline line to the secondary leg
            g16 = {"error": repr(e)}

    # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (committed summary), never from this run
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic_pmc.json")) as f:
            traffic = json.load(f)["k_accumulate_hbm_bytes_per_launch"]["uncorrected"] if args.log2n == 20 else None
    except Exception:
        traffic = None
    result = {
        "metric": "G1 MSM scalar-muls/sec at 2^%d bases per GPU (BLS12-381)" % args.log2n,
        "value": value, "unit": "scalar-muls/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32 limbs (384-bit Montgomery integer)", "data": "synthetic",
        "config": {"workload": "Batched G1 Pippenger MSM, 2^%d random bases/scalars per GPU, bases device-resident with window multiples" % args.log2n,
                   "terms_per_gpu": n, "scalar_bits": 255, "scalar_dist": args.scalar_dist, "sharding": "index range per rank; all_gather of 168-B partial sums" if world > 1 else "none",
                   "bases_setup_s": round(setup_s, 3), "msms_in_flight": DEPTH,
                   "single_msm_latency_ms": round(latency_ms, 3) if world == 1 else None},
        "roofline": {"bound": "hbm", "kernel": L.zkt_last_kernel_name().decode(),
                     "achieved": MSM_BYTES_PER_TERM * n / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": MSM_BYTES_PER_TERM * n / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_note": "FETCH_SIZE+WRITE_SIZE per launch from profiles/r01_hbm_traffic_pmc.json; 13x the algorithmic bytes by design: every term is gathered once per window from the resident window-multiple table (DESIGN.md §4)",
                     "kernel_ms": k_ms, "algorithmic_bytes_per_launch": MSM_BYTES_PER_TERM * n,
                     "note": "integer-VALU bound by construction (SURVEY §8d); see DESIGN.md for the VALU roofline",
                     # the roof that actually bounds the kernel: VALU instruction issue.  5.45 k instructions per bucket add (ISA count of the
                     # k_accumulate hot loop), nwin*n adds per launch; peak = 256 CU x 4 SIMD x 64 lanes / 4 clk at 2.4 GHz.
                     "valu": {"achieved": VALU_INSTR_PER_ADD * NWIN_2P20 * n / (k_ms * 1e-3) / 1e12 if args.log2n == 20 else None, "peak": 39.3, "unit": "T lane-instr/s",
                              "frac": VALU_INSTR_PER_ADD * NWIN_2P20 * n / (k_ms * 1e-3) / 1e12 / 39.3 if args.log2n == 20 else None,
                              "note": "v_mad_u64_u32 (66 % of the mix) issues every ~5 clk, not 4 (profiles/r01_mad_issue_latency.txt): the reachable roof for this mix is ~31-33 T/s"}},
    }

    # parity check of the timed configuration at full size, by linearity: bases are k_i*G, so the MSM over all ranks
    # must equal (sum_i k_i s_i mod r)*G — python integers only, independent of the HIP path
    from zkt_testlib import limbs_to_int
    hk = d_k.cpu().numpy().view(np.uint64)
    tot = 0
    for a, b in zip(hk, h_scalars):
        tot += limbs_to_int(a) * limbs_to_int(b)
    tot %= R_MOD
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, tot)
        tot = sum(parts) % R_MOD
    if rank == 0:
        from zkt_testlib import py_g1_mul, g1_arr
        want = g1_arr([py_g1_mul(G1_GEN, tot)])          # plain python-integer affine arithmetic: independent of the HIP path and of oracle/
        result["config"]["full_size_check"] = "ok" if (want == out).all() else "MISMATCH"

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
            torch.cuda.synchronize(); t0 = time.perf_counter()
            zk.check(L.zkt_tate_batch_dev(vp(d_p), vp(d_q), vp(d_e), m, sp))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            result["pairing"] = {"metric": "Tate pairings/sec", "value": m / dt, "batch": m, "kernel_ms": L.zkt_last_kernel_ms(),
                                 "hbm_achieved_GBs": PAIRING_BYTES * m / dt / 1e9, "hbm_frac": PAIRING_BYTES * m / dt / 1e9 / HBM_PEAK_GBS,
                                 # VALU issue roof: 23.13 M instructions per pairing lane (SQ_INSTS_VALU / SQ_WAVES of k_tate, rocprofv3 --pmc)
                                 "valu": {"achieved": PAIRING_VALU_INSTR * m / dt / 1e12, "peak": 39.3, "unit": "T lane-instr/s", "frac": PAIRING_VALU_INSTR * m / dt / 1e12 / 39.3,
                                          "note": "one wave per SIMD (512 registers per lane): a single wave issues a v_mad_u64_u32 every ~10 clk, other VALU every ~4-5 clk; SQ_WAIT_ANY is 8.5 % of the wave's cycles"}}

        if g16 is not None:
            result["groth16"] = g16

        # CPU baseline: the oracle (faithful restatement of the reference algorithm) on a bounded sample
        if world == 1 and not args.no_cpu:
            from zkt_testlib import oracle, ptr
            O = oracle()
            cores = min(os.cpu_count() or 1, 16)      # the GPU box's CPU share for one GPU is 16 cores
            m = 2048 * cores                           # ~10 s of CPU work on 16 threads
            pts = d_bases[:m].cpu().numpy().view(np.uint64).copy(); sc = h_scalars[:m].copy()
            tmp = np.zeros_like(pts); acc = np.zeros((1, 13), dtype=np.uint64); acc[0, 12] = 1
            t0 = time.perf_counter()
            assert O.zkto_g1_mul_batch(ptr(pts), ptr(sc), 4, ptr(tmp), m, cores) == 0      # n scalar-muls (macros.rs:1-32) on all cores
            for i in range(m):                                                             # n sequential affine adds (polynomial.rs:277-279)
                O.zkto_g1_add_batch(ptr(acc), ptr(tmp[i:i + 1]), ptr(acc), 1)
            dt = time.perf_counter() - t0
            result["cpu_baseline"] = {"value": m / dt, "unit": "scalar-muls/s", "cores": cores, "kind": "port",
                                      "sample": "%d-term MSM by the oracle's reference algorithm (per-term double-and-add on %d threads + sequential affine adds), %.1f s" % (m, cores, dt)}
            if "pairing" in result:                    # the reference's tate(): textbook Miller loop + 4314-bit exponentiation, eight pairings per thread
                mp = 8 * cores
                pp = d_bases[:mp].cpu().numpy().view(np.uint64).copy(); qq = d_q[:mp].cpu().numpy().view(np.uint64).copy()
                oo = np.zeros((mp, 72), dtype=np.uint64); idx = ctypes.c_size_t(0)
                t0 = time.perf_counter()
                assert O.zkto_pairing_batch(3, ptr(pp), ptr(qq), ptr(oo), mp, cores, ctypes.byref(idx)) == 0
                dtp = time.perf_counter() - t0
                same = bool((oo == d_e[:mp].cpu().numpy().view(np.uint64)).all())
                result["pairing"]["cpu_baseline"] = {"value": mp / dtp, "unit": "pairings/s", "cores": cores, "kind": "port",
                                                     "sample": "%d Tate pairings by the oracle's reference algorithm on %d threads, %.1f s" % (mp, cores, dtp),
                                                     "matches_gpu_bits": same}
        print(json.dumps(result))
    L.zkt_g1_bases_free(h)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
