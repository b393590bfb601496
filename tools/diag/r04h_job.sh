#!/bin/bash
# round 4, job h: macro variants of the pairing objects under the adopted scheduler (F: Fq6 products as calls, G: Fq2 products as calls, H: four-scan Fq2 product,
# I/J/K: the Fq12-level routines INLINED into their callers with Fq2 / Fq6 / no products as calls)
set -o pipefail
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O; Z=$PWD/zk-toolkit_amd
run() { ZKT_LIB_PATH=$Z/libzkt_hip$1.so timeout -k 10 200 python3 $2 2>&1 | grep -v "^[WEI]2026\|amdgpu.ids" | tail -${3:-1} | sed "s/^/[$1] /"; }
{
for v in "" _tv4_F _tv4_G _tv4_H _tv4_I _tv4_J _tv4_K; do run "$v" "tools/bench_pairing.py 65536"; done
for v in "" _tv4_F _tv4_G _tv4_H; do run "$v" "tools/bench_g16_batch_verify.py 65536" 1; done
} 2>&1 | tee $O/r04h_macro_sweep.txt
echo done
