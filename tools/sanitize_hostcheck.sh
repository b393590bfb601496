#!/bin/bash
# Host build of the kernel math (csrc/hostcheck.cpp: the same fp.h / tower.h / curve.h / pairing.h the kernels compile) under
# UndefinedBehaviorSanitizer and AddressSanitizer, driven by tests/test_hostcheck.py.  GPU sanitizers are not available on the
# MI355X pool, so this is the sanitizer coverage of the device headers.  Usage: tools/sanitize_hostcheck.sh
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
RT=$(dirname "$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1)")
# both Fq2 products: the default (four product scans) and -DZKT_FQ2_KARATSUBA (three), which the pairing objects are compiled with
for variant in "" "-DZKT_FQ2_KARATSUBA"; do
for san in undefined address; do
  so=/tmp/libzkt_hostcheck_$san.so
  hipcc -x hip --cuda-host-only -O1 -g -std=c++17 -fPIC -shared -fsanitize=$san -fno-sanitize-recover=all -Wno-option-ignored $variant -o "$so" "$ROOT/zk-toolkit_amd/csrc/hostcheck.cpp"
  rt="$RT/libclang_rt.$([ $san = undefined ] && echo ubsan_standalone || echo asan)-x86_64.so"
  echo "== -fsanitize=$san $variant"
  (cd "$ROOT" && LD_PRELOAD="$rt" ASAN_OPTIONS=detect_leaks=0 ZKT_HOSTCHECK_SO="$so" python -m pytest tests/test_hostcheck.py -x -q)
done
done
