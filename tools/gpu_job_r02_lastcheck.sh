# last check of the round: GPU suite and smoke() on the final in-tree build
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/lastcheck_tests.log 2>&1 || { tail -30 gpurun_out/lastcheck_tests.log; echo "tests FAILED"; exit 1; }
tail -2 gpurun_out/lastcheck_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 || exit 1
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 > gpurun_out/lastcheck_bench.json 2> gpurun_out/lastcheck_bench.err || { tail gpurun_out/lastcheck_bench.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/lastcheck_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['full_size_check'], d['groth16']['verifies'], d['bulletproofs']['accepts'])"
