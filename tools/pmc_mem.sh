#!/bin/bash
# Development aid: memory-pipeline counters (L1 stalls, latencies, address translation) of
# the search kernels, one rocprofv3 --pmc pass per group.
# usage: tools/pmc_mem.sh <tag> <time_passes.py arguments...>
set -e
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
ARGS=("$@")
run() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 tools/time_passes.py "${ARGS[@]}" > $out/$name.log 2>&1
}
run m1 TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
run m2 TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum
run m3 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum
python3 tools/pmc_summary.py $out
