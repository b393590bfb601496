# kernel statistics of the protocol-level bench (round 2): BLS, Groth16 batch verification, Bulletproofs, Pinocchio
set -x
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_protocols -- python3 tools/bench_protocols.py > gpurun_out/prof_protocols.json 2> gpurun_out/prof_protocols.err || { tail gpurun_out/prof_protocols.err; exit 1; }
ls gpurun_out/prof_protocols/*/ | head
