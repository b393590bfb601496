#!/bin/bash
# GPU job of the batched-affine go/no-go (tools/ubench/affine_round.hip): timings, then counter passes (their own runs)
set -o pipefail
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 400 ./build/affine_round > $O/r04a_affine_round.txt 2>&1 || { tail -5 $O/r04a_affine_round.txt; exit 1; }
cat $O/r04a_affine_round.txt
for ARGS in "0 1 64" "0 2 32"; do
  T=$(echo $ARGS | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/prof_r04a_sq_$T --output-format csv -- ./build/affine_round $ARGS > $O/r04a_sq_$T.log 2>&1 || { tail -5 $O/r04a_sq_$T.log; exit 1; }
  for CNT in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $CNT -d $O/prof_r04a_${CNT}_$T --output-format csv -- ./build/affine_round $ARGS > $O/r04a_${CNT}_$T.log 2>&1 || { tail -5 $O/r04a_${CNT}_$T.log; exit 1; }
  done
done
echo counters ok
