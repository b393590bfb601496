#!/bin/bash
# One parameterised GPU job (replaces the one-off tools/gpu_job_rNN_*.sh of earlier rounds).  Run through gpurun:
#   gpurun --timeout 900 -- 'bash tools/gpu_job.sh <tag> <step> [<step> ...]'
# Every step writes under gpurun_out/<tag>_*; a failed step ends the job (no further GPU step after a failure or a timeout).
# Steps:
#   tests[:expr]      pytest -m gpu (optionally -k expr)
#   bench             python3 bench.py  -> <tag>_bench.json
#   bench_stats       rocprofv3 --kernel-trace --stats of bench.py (MSM + pairing legs) -> prof_<tag>_stats/
#   msm_latency       tools/bench_msm_latency.py under rocprofv3 --kernel-trace -> <tag>_msm_latency_kernel_trace.txt
#   groth16_stats     tools/bench_groth16.py at 2^20 under rocprofv3 --kernel-trace --stats -> prof_<tag>_g16/ + timeline
#   g16_shard         one rank's share of a 2^20-constraint proof sharded over 8 GPUs (ranks 0, 3, 7, a fresh process each) + the kernel timeline of one share
#   protocols         tools/bench_protocols.py -> <tag>_protocols.json
#   protocols_stats   the same under rocprofv3 --kernel-trace --stats
#   bp                tools/bench_bp.py (range proof / inner-product argument)
#   pmc_acc | pmc_hbm | pmc_tate | pmc_tate_hbm   counter passes (their own runs: no trace domains beside --pmc)
#   pmc_g2            SQ counters, then FETCH_SIZE / WRITE_SIZE passes over tools/bench_g2_msm.py 20 4 (k_accumulate_g2_pair and the Fq2 reduce kernels)
#   ubench            ./build/valu_roof (tools/ubench/valu_roof.hip, built by hipcc beforehand) -> <tag>_valu_ubench.txt
#   verify_latency    tools/bench_verify_latency.py (unprepared / prepared key, calls 1..5; batch sizes)
#   pmc_verify_hbm    FETCH_SIZE / WRITE_SIZE passes over tools/bench_g16_batch_verify.py (the 63-step deciding kernels)
#   rehearsal         two ranks on one card through the callback transport (bench.py --gpus 2, gloo)
#   small_circuits    Groth16 proofs at 2^12 / 2^16 / 2^18 constraints and the Pinocchio latencies -> <tag>_groth16_small_circuits.txt
#   window_sweep      tools/diag/msm_window_sweep.py: forced window widths 12..18 at 2^10..2^18 terms (G1), 14..18 (G2) beside the plan's own table -> <tag>_msm_window_sweep_now.txt
#   verify_timeline   kernel timeline of the first verifications against unseen keys (rocprofv3 --kernel-trace of tools/bench_verify_latency.py) -> <tag>_verify_first_sight_timeline.txt
#   msm_skew          tools/diag/msm_skew.py 14 17 18 20: resident G1 MSM on scalars of 1 / 4 / 8 / 255 bits -> <tag>_msm_small_valued_scalars.txt
#   py:<script> [..]  python3 <script> -> <tag>_<script>.log
#   cmd:<shell>       any shell command (quoted)
set -o pipefail
cd /root/repo
export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out
mkdir -p $O
fail() { echo "STEP FAILED: $1"; tail -25 "$2" 2>/dev/null; exit 1; }
for STEP in "$@"; do
  echo "=== step $STEP ($(date +%H:%M:%S))"
  case "$STEP" in
    tests*) K="${STEP#tests}"; K="${K#:}"
      timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=10 ${K:+-k "$K"} > $O/${TAG}_gputests.log 2>&1 || fail tests $O/${TAG}_gputests.log
      tail -14 $O/${TAG}_gputests.log ;;
    bench) timeout -k 10 900 python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || fail bench $O/${TAG}_bench.err
      python3 -c "import json,sys; d=json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step')}, d.get('roofline'), d['config'].get('single_msm_latency_ms'), d.get('pairing',{}).get('value'), d.get('groth16',{}).get('value'), d.get('bulletproofs'))" ;;
    bench_stats) timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_stats -- python3 bench.py --groth16-log2n 0 --g2-log2n 0 > $O/${TAG}_bench_msm_pairing.json 2> $O/prof_${TAG}_stats.err || fail bench_stats $O/prof_${TAG}_stats.err ;;
    msm_latency) timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_${TAG}_lat -- python3 tools/bench_msm_latency.py > $O/${TAG}_msm_latency.log 2>&1 || fail msm_latency $O/${TAG}_msm_latency.log
      grep "latency" $O/${TAG}_msm_latency.log
      python3 tools/trace_summary.py $(ls $O/prof_${TAG}_lat/*/*kernel_trace.csv | head -1) > $O/${TAG}_msm_latency_kernel_trace.txt 2>&1 || true
      cat $O/${TAG}_msm_latency_kernel_trace.txt | cut -c1-150 ;;
    groth16_stats) timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_g16 -- python3 tools/bench_groth16.py --log-n 20 --proofs 8 > $O/prof_${TAG}_g16.log 2>&1 || fail groth16_stats $O/prof_${TAG}_g16.log
      grep -v "^[WEI]2026" $O/prof_${TAG}_g16.log | tail -4
      cp $(ls $O/prof_${TAG}_g16/*/*kernel_stats.csv | head -1) $O/${TAG}_groth16_2p20_kernel_stats.csv
      python3 tools/trace_timeline.py $(ls $O/prof_${TAG}_g16/*/*kernel_trace.csv | head -1) 3 140 > $O/${TAG}_g16_timeline.txt 2>&1 || true ;;
    g16_shard) : > $O/${TAG}_g16_shard.txt
      for RK in 0 3 7; do
        timeout -k 10 300 python3 tools/bench_groth16.py --log-n 20 --proofs 8 --shard-of 8 --rank $RK > $O/${TAG}_g16_shard_$RK.log 2>&1 || fail "g16_shard $RK" $O/${TAG}_g16_shard_$RK.log
        grep "^rank" $O/${TAG}_g16_shard_$RK.log >> $O/${TAG}_g16_shard.txt
      done
      timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_${TAG}_g16_shard -- python3 tools/bench_groth16.py --log-n 20 --proofs 3 --shard-of 8 --rank 3 > $O/prof_${TAG}_g16_shard.log 2>&1 || fail g16_shard $O/prof_${TAG}_g16_shard.log
      echo "--- kernel timeline of one share (rank 3; rocprofv3 --kernel-trace, from the proof's first kernel):" >> $O/${TAG}_g16_shard.txt
      python3 tools/trace_timeline.py $(ls -t $O/prof_${TAG}_g16_shard/*/*kernel_trace.csv | head -1) 3 200 k_to_mont | grep -v "k_scan\|k_zero" >> $O/${TAG}_g16_shard.txt 2>&1 || true
      head -4 $O/${TAG}_g16_shard.txt ;;
    protocols) timeout -k 10 900 python3 tools/bench_protocols.py > $O/${TAG}_protocols.json 2> $O/${TAG}_protocols.err || fail protocols $O/${TAG}_protocols.err
      cat $O/${TAG}_protocols.json | cut -c1-1500 ;;
    protocols_stats) timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_protocols -- python3 tools/bench_protocols.py > $O/${TAG}_protocols_b.json 2> $O/prof_${TAG}_protocols.err || fail protocols_stats $O/prof_${TAG}_protocols.err ;;
    bp) timeout -k 10 600 python3 tools/bench_bp.py > $O/${TAG}_bp.log 2>&1 || fail bp $O/${TAG}_bp.log
      tail -12 $O/${TAG}_bp.log ;;
    pmc_acc) timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $O/prof_${TAG}_sq_acc --output-format csv -- python3 bench.py --no-cpu --no-bulletproofs --steps 8 --warmup 2 --pairings 0 --groth16-log2n 0 --g2-log2n 0 > $O/prof_${TAG}_sq_acc.log 2>&1 || fail pmc_acc $O/prof_${TAG}_sq_acc.log ;;
    pmc_hbm) for CNT in FETCH_SIZE WRITE_SIZE; do      # one counter per pass: together they exceed what the hardware collects at once (rocprofv3 aborts)
        timeout -k 10 300 rocprofv3 --pmc $CNT -d $O/prof_${TAG}_hbm_$CNT --output-format csv -- python3 bench.py --no-cpu --no-bulletproofs --steps 8 --warmup 2 --pairings 0 --groth16-log2n 0 --g2-log2n 0 > $O/prof_${TAG}_hbm_$CNT.log 2>&1 || fail "pmc_hbm $CNT" $O/prof_${TAG}_hbm_$CNT.log
      done ;;
    pmc_g2) timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/prof_${TAG}_sq_g2 --output-format csv -- python3 tools/bench_g2_msm.py 20 4 > $O/prof_${TAG}_sq_g2.log 2>&1 || fail pmc_g2 $O/prof_${TAG}_sq_g2.log
      for CNT in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 300 rocprofv3 --pmc $CNT -d $O/prof_${TAG}_hbm_g2_$CNT --output-format csv -- python3 tools/bench_g2_msm.py 20 4 > $O/prof_${TAG}_hbm_g2_$CNT.log 2>&1 || fail "pmc_g2 $CNT" $O/prof_${TAG}_hbm_g2_$CNT.log
      done ;;
    ubench) timeout -k 10 300 ./build/valu_roof > $O/${TAG}_valu_ubench.txt 2>&1 || fail ubench $O/${TAG}_valu_ubench.txt
      grep "v_mad_u64_u32\|field mix" $O/${TAG}_valu_ubench.txt | head -8 ;;
    verify_latency) timeout -k 10 300 python3 tools/bench_verify_latency.py 2>&1 | grep -v "^[WEI]2026\|amdgpu.ids" > $O/${TAG}_verify_latency.txt || fail verify_latency $O/${TAG}_verify_latency.txt
      head -3 $O/${TAG}_verify_latency.txt ;;
    pmc_tate) timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/prof_${TAG}_sq_tate --output-format csv -- python3 tools/bench_pairing.py 65536 > $O/prof_${TAG}_sq_tate.log 2>&1 || fail pmc_tate $O/prof_${TAG}_sq_tate.log ;;
    pmc_tate_hbm) for CNT in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 300 rocprofv3 --pmc $CNT -d $O/prof_${TAG}_hbm_tate_$CNT --output-format csv -- python3 tools/bench_pairing.py 65536 > $O/prof_${TAG}_hbm_tate_$CNT.log 2>&1 || fail "pmc_tate_hbm $CNT" $O/prof_${TAG}_hbm_tate_$CNT.log
      done ;;
    pmc_verify_hbm) for CNT in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 300 rocprofv3 --pmc $CNT -d $O/prof_${TAG}_hbm_verify_$CNT --output-format csv -- python3 tools/bench_g16_batch_verify.py 65536 > $O/prof_${TAG}_hbm_verify_$CNT.log 2>&1 || fail "pmc_verify_hbm $CNT" $O/prof_${TAG}_hbm_verify_$CNT.log
      done ;;
    rehearsal) ZKT_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --groth16-log2n 16 --groth16-proofs 3 --pairings 4096 --no-cpu > $O/${TAG}_bench_rehearsal_2ranks_1gpu.json 2> $O/${TAG}_rehearsal.err || fail rehearsal $O/${TAG}_rehearsal.err ;;
    small_circuits) : > $O/${TAG}_groth16_small_circuits.txt
      for LN in 12 16 18; do timeout -k 10 300 python3 tools/bench_groth16.py --log-n $LN --proofs 10 2>&1 | grep "^prove" | sed "s/^/2^$LN /" >> $O/${TAG}_groth16_small_circuits.txt || fail "small_circuits $LN" $O/${TAG}_groth16_small_circuits.txt; done
      timeout -k 10 300 python3 tools/bench_pinocchio.py 2>&1 | grep -v "^[WEI]2026\|amdgpu.ids" | tail -4 >> $O/${TAG}_groth16_small_circuits.txt || fail small_circuits $O/${TAG}_groth16_small_circuits.txt
      cat $O/${TAG}_groth16_small_circuits.txt ;;
    window_sweep) : > $O/${TAG}_msm_window_sweep_now.txt
      timeout -k 10 300 python3 tools/diag/msm_window_sweep.py g1 4 8 10 12 14 16 17 18 19 2>&1 | grep "^c " >> $O/${TAG}_msm_window_sweep_now.txt || fail window_sweep $O/${TAG}_msm_window_sweep_now.txt
      for C in 12 14 16 17 18; do ZKT_MSM_C=$C timeout -k 10 300 python3 tools/diag/msm_window_sweep.py g1 10 12 14 16 17 18 2>&1 | grep "^c " >> $O/${TAG}_msm_window_sweep_now.txt || fail "window_sweep $C" $O/${TAG}_msm_window_sweep_now.txt; done
      for C in 14 16 18; do ZKT_MSM_C=$C timeout -k 10 300 python3 tools/diag/msm_window_sweep.py g2 14 17 18 2>&1 | grep "^c " >> $O/${TAG}_msm_window_sweep_now.txt || fail "window_sweep g2 $C" $O/${TAG}_msm_window_sweep_now.txt; done
      tail -5 $O/${TAG}_msm_window_sweep_now.txt ;;
    verify_timeline) timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_${TAG}_vl -- python3 tools/bench_verify_latency.py > $O/prof_${TAG}_vl.log 2>&1 || fail verify_timeline $O/prof_${TAG}_vl.log
      python3 tools/diag/verify_trace.py $(ls -t $O/prof_${TAG}_vl/*/*kernel_trace.csv | head -1) 60 > $O/${TAG}_verify_first_sight_timeline.txt 2>&1 || true
      head -14 $O/${TAG}_verify_first_sight_timeline.txt ;;
    msm_skew) timeout -k 10 400 python3 tools/diag/msm_skew.py 14 17 18 20 2>&1 | grep "^g1" > $O/${TAG}_msm_small_valued_scalars.txt || fail msm_skew $O/${TAG}_msm_small_valued_scalars.txt
      cat $O/${TAG}_msm_small_valued_scalars.txt ;;
    py:*) S="${STEP#py:}"; N=$(basename ${S%% *} .py)
      timeout -k 10 900 python3 $S > $O/${TAG}_${N}.log 2>&1 || fail "$STEP" $O/${TAG}_${N}.log
      grep -v "^[WEI]2026" $O/${TAG}_${N}.log | tail -25 ;;
    cmd:*) timeout -k 10 900 bash -c "${STEP#cmd:}" || { echo "STEP FAILED: $STEP"; exit 1; } ;;
    *) echo "unknown step $STEP"; exit 2 ;;
  esac
done
echo all steps ok
