#!/bin/bash
# round 4, job j: one call per step of the 63-step product loop — verification / signature rates, then the deciding entry points' tests
set -o pipefail
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
{ timeout -k 10 200 python3 tools/bench_g16_batch_verify.py 65536 2>&1 | grep "proofs,"
  timeout -k 10 200 python3 tools/bench_pairing.py 65536 2>&1 | grep batch; } | tee $O/r04j_step_lines.txt
echo "=== tests"; timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "verify or signature or pinocchio_vs or pairing_product or verification or outside" > $O/r04j_tests.log 2>&1; tail -4 $O/r04j_tests.log
echo "=== protocols"; timeout -k 10 600 python3 tools/bench_protocols.py > $O/r04j_protocols.json 2> $O/r04j_protocols.err; python3 -c "
import json; d=json.load(open('$O/r04j_protocols.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, {a:b for a,b in v.items() if isinstance(b,(int,float))})
" 2>/dev/null | cut -c1-300
echo done
