# host side of one range proof: HIP runtime API trace beside the kernel trace (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --hip-runtime-trace --kernel-trace --stats --output-format csv -d gpurun_out/prof_bp_hip -- python3 tools/bench_bp_rp_only.py 0 > gpurun_out/prof_bp_hip.log 2>&1 || { tail gpurun_out/prof_bp_hip.log; exit 1; }
grep "range proof" gpurun_out/prof_bp_hip.log | tail -2
ls gpurun_out/prof_bp_hip/*/
