# are the reduce kernels' 150-600 MB "use-once" scratch allocations (frames of 288-600 B/lane) what staggers concurrent small MSMs?  Same runs with the
# runtime's single-queue retention limit raised so that those allocations are kept (round 2)
set -x
cd /root/repo
export TMPDIR=/tmp
for lim in default 2000000000; do
  if [ $lim = default ]; then unset HSA_SCRATCH_SINGLE_LIMIT; else export HSA_SCRATCH_SINGLE_LIMIT=$lim; fi
  echo "== HSA_SCRATCH_SINGLE_LIMIT=$lim"
  timeout -k 10 300 python3 tools/bench_bp_rp_only.py 0 2>&1 | grep "range proof" | tail -2 || exit 1
  timeout -k 10 300 python3 tools/bench_msm_latency.py 2>&1 | grep "single MSM" || exit 1
  timeout -k 10 300 python3 tools/bench_pinocchio.py --reps 4 2>&1 | tail -1 || exit 1
done
