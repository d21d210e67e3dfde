#!/bin/bash
# Usage (on the GPU box): bash tools/gpu_pmc_mem.sh <tag> [driver args...]   -- memory-pipeline counters of the traversal kernels
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --pmc TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE -d $OUT/pmc4 -o pmc4 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py "$@" > $OUT/pmc4.log 2>&1 || exit 2
timeout -s KILL 180 rocprofv3 --pmc TCP_TCP_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_HIT_sum TA_FLAT_READ_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum -d $OUT/pmc5 -o pmc5 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py "$@" > $OUT/pmc5.log 2>&1 || exit 3
timeout -s KILL 180 rocprofv3 --pmc TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCC_REQ_sum TCC_READ_sum TCC_BUSY_sum TCC_TAG_STALL_sum -d $OUT/pmc6 -o pmc6 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py "$@" > $OUT/pmc6.log 2>&1 || exit 4
