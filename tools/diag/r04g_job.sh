#!/bin/bash
# round 4, job g: third scheduler sweep on the two pairing objects (baseline = max-memory-clause, adopted), then the pairing-family tests on the adopted build
set -o pipefail
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O; Z=$PWD/zk-toolkit_amd
run() { ZKT_LIB_PATH=$Z/libzkt_hip$1.so timeout -k 10 300 python3 $2 2>&1 | grep -v "^[WEI]2026\|amdgpu.ids" | tail -${3:-1} | sed "s/^/[$1] /"; }
{
for v in "" _tv3_B _tv3_C _tv3_D _tv3_E; do run "$v" "tools/bench_pairing.py 65536"; run "$v" "tools/bench_g16_batch_verify.py 65536" 1; done
} 2>&1 | tee $O/r04g_sched_sweep3.txt
echo "=== tests (adopted build)"; timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "pairing or tate or weil or miller or verify or signature or pinocchio or dpairing or config3 or verification" > $O/r04g_tests.log 2>&1; tail -4 $O/r04g_tests.log
echo done
