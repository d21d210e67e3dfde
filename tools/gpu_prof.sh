#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/gpu_prof.sh <tag> [driver args...]
# Writes rocprofv3 kernel-trace stats and four hardware-counter passes under gpurun_out/prof_<tag>/.
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py "$@" > $OUT/kt.log 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc1 -o pmc1 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py "$@" > $OUT/pmc1.log 2>&1 || exit 2
timeout -s KILL 180 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 -d $OUT/pmc2 -o pmc2 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py "$@" > $OUT/pmc2.log 2>&1 || exit 3
timeout -s KILL 180 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE -d $OUT/pmc3 -o pmc3 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py "$@" > $OUT/pmc3.log 2>&1 || exit 4
timeout -s KILL 180 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE -d $OUT/pmc4 -o pmc4 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py "$@" > $OUT/pmc4.log 2>&1 || exit 5
