# kernel durations after the lane-activity change: window joins, fixed tables, guards, trees (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lanes -- python3 tools/bench_pinocchio.py --reps 3 > gpurun_out/prof_lanes.log 2>&1 || { tail gpurun_out/prof_lanes.log; exit 1; }
f=$(ls gpurun_out/prof_lanes/*/*kernel_stats.csv | head -1)
grep "k_join_windows\|k_fixed_table\|k_short_loop_guards\|k_pin_sums\|k_generator\|k_dproduct\|k_combine\|k_marginals" $f | sed 's/(.*)"/"/' | cut -c1-140
