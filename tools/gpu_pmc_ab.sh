#!/bin/bash
# Usage (on the GPU box): bash tools/gpu_pmc_ab.sh <workload> <spp> <variant>...   -- wait / LDS / memory-pipeline counters of
# the traversal kernels for library variants built by tools/build_variant.sh ("main" = the regular build)
set -u
WL=$1; SPP=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" != main ]; then export PT_HIP_LIB_DIR=$GRAFT_REPO_ROOT/build/var_$v; else unset PT_HIP_LIB_DIR; fi
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ab_$v
  rm -rf $OUT; mkdir -p $OUT
  D="python3 $GRAFT_REPO_ROOT/tools/prof_driver.py --workload $WL --spp $SPP --traversal fast"
  timeout -s KILL 180 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $OUT/pmc1 -o pmc1 -- $D > $OUT/pmc1.log 2>&1 || exit 2
  timeout -s KILL 180 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d $OUT/pmc2 -o pmc2 -- $D > $OUT/pmc2.log 2>&1 || exit 3
  timeout -s KILL 180 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE -d $OUT/pmc3 -o pmc3 -- $D > $OUT/pmc3.log 2>&1 || exit 4
  timeout -s KILL 180 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d $OUT/pmc4 -o pmc4 -- $D > $OUT/pmc4.log 2>&1 || exit 5
  python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT > $GRAFT_REPO_ROOT/gpurun_out/ab_${v}_pmc.txt
  rm -rf $OUT
done
