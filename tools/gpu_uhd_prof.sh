#!/bin/bash
# Usage (on the GPU box): bash tools/gpu_uhd_prof.sh <tag> <workload> <spp>
# Kernel trace + the SQ / TCC / TCP counter passes of one workload (meant for the 49 M triangle, 3840x2160 case); only
# the text summary is kept (the rocprofv3 databases of such a run exceed what a call may bring back).
set -u
TAG=$1; W=$2; SPP=$3
R=$GRAFT_REPO_ROOT
bash tools/gpu_prof.sh $TAG --workload "$W" --spp $SPP --traversal fast || exit 1
bash tools/gpu_pmc_mem.sh $TAG --workload "$W" --spp $SPP --traversal fast || exit 2
python3 tools/prof_summary.py $R/gpurun_out/prof_$TAG > $R/gpurun_out/${TAG}_prof.txt || exit 3
grep -h " rays " $R/gpurun_out/prof_$TAG/kt.log >> $R/gpurun_out/${TAG}_prof.txt
rm -rf $R/gpurun_out/prof_$TAG
