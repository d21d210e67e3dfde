#!/bin/bash
# Usage (on the GPU box): bash tools/gpu_pmc_sq.sh <tag> [driver args...]   -- the two SQ counter passes of gpu_prof.sh only
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc1 -o pmc1 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py "$@" > $OUT/pmc1.log 2>&1 || exit 2
timeout -s KILL 180 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 -d $OUT/pmc2 -o pmc2 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py "$@" > $OUT/pmc2.log 2>&1 || exit 3
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_sq.txt
rm -rf $OUT
