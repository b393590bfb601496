#!/bin/bash
# Fold what a `tools/gpu_job.sh <tag> ...` run left under gpurun_out/ into profiles/<tag>_* (the files the judge reads and bench.py loads):
# the NEWEST counter_collection / kernel_stats CSV of every pass, the bench lines, the traces.  Counter JSONs get the hash of the kernel sources of THIS tree
# (tools/src_hash.py) — run it on the tree the job was sent from.      usage: tools/fold_profiles.sh r03
set -e
cd "$(dirname "$0")/.."
T=$1; O=gpurun_out; P=profiles
newest() { ls -t $1 2>/dev/null | head -1; }
UB=$P/${T}_valu_ubench.txt
[ -f $O/${T}_valu_ubench.txt ] && cp $O/${T}_valu_ubench.txt $UB
BENCHCMD="python3 bench.py --no-cpu --no-bulletproofs --steps 8 --warmup 2 --pairings 0 --groth16-log2n 0"
python3 tools/sq_summary.py $P/${T}_accumulate_sq_counters.json $UB "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -- $BENCHCMD (MI355X, round ${T#r}, final build; tools/gpu_job.sh $T pmc_acc)" "$(newest "$O/prof_${T}_sq_acc/*/*counter_collection.csv")"
python3 tools/pmc_summary.py $P/${T}_hbm_traffic_pmc.json "rocprofv3 --pmc FETCH_SIZE (one pass) and --pmc WRITE_SIZE (another) -- $BENCHCMD (MI355X, round ${T#r}, final build; tools/gpu_job.sh $T pmc_hbm)" "$(newest "$O/prof_${T}_hbm_FETCH_SIZE/*/*counter_collection.csv")" "$(newest "$O/prof_${T}_hbm_WRITE_SIZE/*/*counter_collection.csv")"
python3 tools/sq_summary.py $P/${T}_tate_sq_counters.json $UB "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- python3 tools/bench_pairing.py 65536 (MI355X, round ${T#r}, final build; tools/gpu_job.sh $T pmc_tate)" "$(newest "$O/prof_${T}_sq_tate/*/*counter_collection.csv")"
python3 tools/pmc_summary.py $P/${T}_tate_memory_counters.json "rocprofv3 --pmc FETCH_SIZE (one pass) / --pmc WRITE_SIZE (another) -- python3 tools/bench_pairing.py 65536 (MI355X, round ${T#r}, final build; tools/gpu_job.sh $T pmc_tate_hbm).  Averages per launch at 65,536 pairings." "$(newest "$O/prof_${T}_hbm_tate_FETCH_SIZE/*/*counter_collection.csv")" "$(newest "$O/prof_${T}_hbm_tate_WRITE_SIZE/*/*counter_collection.csv")"
if ls $O/prof_${T}_hbm_verify_FETCH_SIZE/*/*counter_collection.csv > /dev/null 2>&1; then
python3 tools/pmc_summary.py $P/${T}_verify_memory_counters.json "rocprofv3 --pmc FETCH_SIZE (one pass) / --pmc WRITE_SIZE (another) -- python3 tools/bench_g16_batch_verify.py 65536 (MI355X, round ${T#r}, final build; tools/gpu_job.sh $T pmc_verify_hbm)" "$(newest "$O/prof_${T}_hbm_verify_FETCH_SIZE/*/*counter_collection.csv")" "$(newest "$O/prof_${T}_hbm_verify_WRITE_SIZE/*/*counter_collection.csv")"
fi
if ls $O/prof_${T}_sq_g2/*/*counter_collection.csv > /dev/null 2>&1; then
python3 tools/sq_summary.py $P/${T}_g2_msm_sq_counters.json $UB "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- python3 tools/bench_g2_msm.py 20 4 (MI355X, round ${T#r}, final build; tools/gpu_job.sh $T pmc_g2): the G2 accumulate on lane pairs and the Fq2 reduce kernels at 2^20 terms" "$(newest "$O/prof_${T}_sq_g2/*/*counter_collection.csv")" > /dev/null
python3 tools/pmc_summary.py $P/${T}_g2_msm_memory_counters.json "rocprofv3 --pmc FETCH_SIZE (one pass) / --pmc WRITE_SIZE (another) -- python3 tools/bench_g2_msm.py 20 4 (MI355X, round ${T#r}, final build; tools/gpu_job.sh $T pmc_g2)" "$(newest "$O/prof_${T}_hbm_g2_FETCH_SIZE/*/*counter_collection.csv")" "$(newest "$O/prof_${T}_hbm_g2_WRITE_SIZE/*/*counter_collection.csv")" > /dev/null
fi
[ -f $O/${T}_verify_latency.txt ] && cp $O/${T}_verify_latency.txt $P/${T}_verify_latency.txt
cp $O/${T}_bench.json $P/${T}_bench.json
cp $O/${T}_bench_msm_pairing.json $P/${T}_bench_msm_pairing.json
cp "$(newest "$O/prof_${T}_stats/*/*kernel_stats.csv")" $P/${T}_bench_kernel_stats.csv
cp $O/${T}_msm_latency_kernel_trace.txt $P/${T}_msm_latency_kernel_trace.txt
cp $O/${T}_groth16_2p20_kernel_stats.csv $P/${T}_groth16_2p20_kernel_stats.csv
cp $O/${T}_g16_timeline.txt $P/${T}_groth16_timeline.txt
cp $O/${T}_protocols.json $P/${T}_protocols.json
cp "$(newest "$O/prof_${T}_protocols/*/*kernel_stats.csv")" $P/${T}_protocols_kernel_stats.csv
[ -f $O/${T}_g16_shard.txt ] && cp $O/${T}_g16_shard.txt $P/${T}_groth16_shard_of_8.txt
[ -f $O/${T}_bp.log ] && grep -v "amdgpu.ids" $O/${T}_bp.log > $P/${T}_bulletproofs_latency.txt
for X in groth16_small_circuits verify_first_sight_timeline msm_small_valued_scalars; do [ -f $O/${T}_$X.txt ] && cp $O/${T}_$X.txt $P/${T}_$X.txt; done
echo "folded into $P/${T}_*  (kernel sources $(python3 tools/src_hash.py))"
python3 - "$P/${T}_tate_memory_counters.json" <<'PY'
import json, sys
p = sys.argv[1]; d = json.load(open(p)); k = d["kernels"].get("k_tate")
if k:
    gb = (k["FETCH_SIZE_KB_avg_per_launch"] + k["WRITE_SIZE_KB_avg_per_launch"]) * 1024 / 1e9
    d["k_tate_summary"] = {"memory_side_GB_per_launch_lower_bound": gb, "pairings_per_launch": 65536, "algorithmic_bytes_per_launch": 65536 * 864,
                           "traffic_over_algorithmic": gb * 1e9 / (65536 * 864)}
    json.dump(d, open(p, "w"), indent=1)
PY
