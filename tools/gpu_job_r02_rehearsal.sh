# smoke() of __graft_entry__ and the 2-rank rehearsal of bench.py on one GPU (gloo transport for the exchange), round 2
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02_smoke.log 2>&1 || { tail -20 gpurun_out/r02_smoke.log; exit 1; }
tail -2 gpurun_out/r02_smoke.log
ZKT_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --groth16-log2n 16 --groth16-proofs 3 --pairings 4096 --no-cpu > gpurun_out/r02_bench_rehearsal_2ranks_1gpu.json 2> gpurun_out/r02_rehearsal.err || { tail -20 gpurun_out/r02_rehearsal.err; exit 1; }
grep "^{" gpurun_out/r02_bench_rehearsal_2ranks_1gpu.json | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value']/1e6, d.get('pairing_all_gpus'), d.get('strong',{}).get('value'), d['groth16'].get('sharded'))"
