set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_bp -- python3 tools/bench_bp_rp_only.py 0 > gpurun_out/prof_bp.log 2>&1 || { tail gpurun_out/prof_bp.log; exit 1; }
grep "range proof" gpurun_out/prof_bp.log
