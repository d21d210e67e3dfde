#!/bin/bash
# Scratch: full GPU test suite, then the two quick throughput probes.  Usage: bash tools/gpu_quick.sh <tag>
set -o pipefail
tag=${1:-q}
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/${tag}_tests.log 2>&1 || { tail -30 gpurun_out/${tag}_tests.log; exit 1; }
tail -2 gpurun_out/${tag}_tests.log
timeout -k 10 300 python -u tools/gpu_pass.py library-synth 256 128 > gpurun_out/${tag}_lib.log 2>&1 || { tail -5 gpurun_out/${tag}_lib.log; exit 1; }
tail -1 gpurun_out/${tag}_lib.log
timeout -k 10 300 python -u tools/gpu_pass.py veach-mis 256 128 > gpurun_out/${tag}_veach.log 2>&1 || { tail -5 gpurun_out/${tag}_veach.log; exit 1; }
tail -1 gpurun_out/${tag}_veach.log
