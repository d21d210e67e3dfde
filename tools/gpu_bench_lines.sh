#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/gpu_bench_lines.sh <tag>
# The bench lines of a measurement round alone (both headline workloads in both traversal modes, the CPU-runnable case and
# one GPU's view of the tile-split case), for when bench.py's output changed and the kernels did not.
set -u
TAG=$1
O=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 300 python -u bench.py > $O/${TAG}_bench_library_fast.json 2> $O/${TAG}_bench_library_fast.err || exit 2
timeout -k 10 300 python -u bench.py --workload veach-mis > $O/${TAG}_bench_veach_fast.json 2> $O/${TAG}_bench_veach_fast.err || exit 3
timeout -k 10 300 python -u bench.py --workload veach-mis --traversal exact --no-cpu-baseline > $O/${TAG}_bench_veach_exact.json 2> $O/${TAG}_bench_veach_exact.err || exit 4
timeout -k 10 300 python -u bench.py --traversal exact --spp 256 --no-cpu-baseline > $O/${TAG}_bench_library_exact.json 2> $O/${TAG}_bench_library_exact.err || exit 5
timeout -k 10 300 python -u bench.py --workload cornell-box-synth --spp 64 --steps 20 --warmup 2 > $O/${TAG}_bench_cornell64spp_fast.json 2> $O/${TAG}_bench_cornell64spp_fast.err || exit 12
timeout -k 10 400 python -u bench.py --workload bathroom-synth --spp 2048 --steps 2 --warmup 1 > $O/${TAG}_bench_bathroom2048spp_fast.json 2> $O/${TAG}_bench_bathroom2048spp_fast.err || exit 13
