#!/bin/bash
# Scratch: A/B throughput of library variants built by tools/build_variant.sh.  Usage: bash tools/gpu_ab.sh <name>...
set -o pipefail
mkdir -p gpurun_out
for v in "$@"; do
  for w in library-synth veach-mis; do
    PT_HIP_LIB_DIR=$PWD/build/var_$v timeout -k 10 200 python -u tools/gpu_pass.py $w 256 128 > gpurun_out/ab_${v}_$w.log 2>&1 || { tail -5 gpurun_out/ab_${v}_$w.log; exit 1; }
    echo "$v $(tail -1 gpurun_out/ab_${v}_$w.log)"
  done
done
