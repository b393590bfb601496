# round-2 record run: GPU suite, bench line, kernel statistics of the bench, protocol figures, two-rank rehearsal.  A failed step ends the job.
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r02_gputests.log 2>&1 || { tail -30 gpurun_out/r02_gputests.log; echo "tests FAILED"; exit 1; }
tail -3 gpurun_out/r02_gputests.log
timeout -k 10 900 python3 bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err || { tail gpurun_out/r02_bench.err; echo "bench FAILED"; exit 1; }
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --groth16-log2n 0 > gpurun_out/r02_bench_msm_pairing.json 2> gpurun_out/prof_stats.err || { tail gpurun_out/prof_stats.err; echo "stats FAILED"; exit 1; }
timeout -k 10 900 python3 tools/bench_protocols.py > gpurun_out/r02_protocols.json 2> gpurun_out/r02_protocols.err || { tail gpurun_out/r02_protocols.err; echo "protocols FAILED"; exit 1; }
ZKT_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --groth16-log2n 16 --groth16-proofs 3 --pairings 4096 --no-cpu > gpurun_out/r02_bench_rehearsal_2ranks_1gpu.json 2> gpurun_out/r02_rehearsal.err || { tail gpurun_out/r02_rehearsal.err; echo "rehearsal FAILED"; exit 1; }
echo all steps ok
