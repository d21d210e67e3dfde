#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/gpu_round.sh <tag>
# GPU test suite, the bench lines and the rocprofv3 passes of one measurement round; everything lands under gpurun_out/.
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/${TAG}_gputests.log 2>&1 || exit 1
timeout -k 10 300 python -u bench.py > $O/${TAG}_bench_library_fast.json 2> $O/${TAG}_bench_library_fast.err || exit 2
timeout -k 10 300 python -u bench.py --workload veach-mis > $O/${TAG}_bench_veach_fast.json 2> $O/${TAG}_bench_veach_fast.err || exit 3
timeout -k 10 300 python -u bench.py --workload veach-mis --traversal exact --no-cpu-baseline > $O/${TAG}_bench_veach_exact.json 2> $O/${TAG}_bench_veach_exact.err || exit 4
timeout -k 10 300 python -u bench.py --traversal exact --spp 256 --no-cpu-baseline > $O/${TAG}_bench_library_exact.json 2> $O/${TAG}_bench_library_exact.err || exit 5
# the other configurations of the baseline that fit one GPU: the CPU-runnable case and one GPU's view of the tile-split case
timeout -k 10 300 python -u bench.py --workload cornell-box-synth --spp 64 --steps 20 --warmup 2 > $O/${TAG}_bench_cornell64spp_fast.json 2> $O/${TAG}_bench_cornell64spp_fast.err || exit 12
timeout -k 10 400 python -u bench.py --workload bathroom-synth --spp 2048 --steps 2 --warmup 1 > $O/${TAG}_bench_bathroom2048spp_fast.json 2> $O/${TAG}_bench_bathroom2048spp_fast.err || exit 13
# kernel trace of the bench command itself (same workload, same pass size; fewer steps)
mkdir -p $O/prof_${TAG}_bench
(cd /tmp && export TMPDIR=/tmp && timeout -s KILL 280 rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_bench/kt -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_${TAG}_bench/kt.log 2>&1) || exit 6
python3 tools/prof_summary.py $O/prof_${TAG}_bench > $O/${TAG}_prof_bench_library_fast.txt || exit 7
bash tools/gpu_prof.sh ${TAG}_library --workload library-synth --spp 512 --traversal fast || exit 8
python3 tools/prof_summary.py $O/prof_${TAG}_library > $O/${TAG}_prof_library_fast.txt || exit 9
bash tools/gpu_prof.sh ${TAG}_veach --workload veach-mis --spp 512 --traversal fast || exit 10
python3 tools/prof_summary.py $O/prof_${TAG}_veach > $O/${TAG}_prof_veach_fast.txt || exit 11
# the rocprofv3 databases are not needed once summarised (and a call may bring back 64 MiB at most)
rm -rf $O/prof_${TAG}_bench $O/prof_${TAG}_library $O/prof_${TAG}_veach
