#!/bin/bash
# Development aid: hardware counters of the search kernels, one rocprofv3 --pmc pass per
# counter group (at most 8 SQ_ / 4 TCC_ / 4 TCP_ counters per pass).
# usage: tools/pmc.sh <tag> <time_passes.py arguments...>
set -e
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
run() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 tools/time_passes.py "${ARGS[@]}" > $out/$name.log 2>&1
}
ARGS=("$@")
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_FMA_F64 SQ_LDS_BANK_CONFLICT
run tc1 TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
run tc2 FETCH_SIZE
run tc3 WRITE_SIZE
python3 tools/pmc_summary.py $out
