# the bench line and its kernel statistics (round 2), after profiles/ was refreshed
set -x
cd /root/repo
export TMPDIR=/tmp
python3 bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --groth16-log2n 0 > gpurun_out/r02_bench_msm_pairing.json 2> gpurun_out/prof_stats.err || exit 1
echo done
