#!/bin/bash
# Usage (on the GPU box, after tools/install_round.py has refreshed profiles/traffic.json): bash tools/gpu_round_post.sh <tag>
# The two headline bench lines against the refreshed counters (so that they do not print "stale"), then the kernel trace,
# the counter passes and the throughput of the 49 M triangle / 3840x2160 case.  Afterwards, here:
#   cp gpurun_out/<tag>_bench_{library,veach}_fast.json profiles/r3_<letter>_bench_{library,veach}_fast.json
#   python tools/uhd_summary.py <tag>_uhd <letter> <Mrays/s of the last line of gpurun_out/<tag>_uhd_pass.log>
set -u
TAG=$1
O=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 300 python -u bench.py > $O/${TAG}_bench_library_fast.json 2> $O/${TAG}_bench_library_fast.err || exit 1
timeout -k 10 300 python -u bench.py --workload veach-mis > $O/${TAG}_bench_veach_fast.json 2> $O/${TAG}_bench_veach_fast.err || exit 2
bash tools/gpu_uhd_prof.sh ${TAG}_uhd "library-synth@44:3840x2160" 16 || exit 3
timeout -k 10 300 python -u tools/gpu_pass.py "library-synth@44:3840x2160" 64 64 > $O/${TAG}_uhd_pass.log 2>&1 || exit 4
tail -1 $O/${TAG}_uhd_pass.log
