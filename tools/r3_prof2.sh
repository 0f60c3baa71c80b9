#!/bin/bash
# round-3 counters: memory-side traffic per kernel and SQ counters per kernel of the bench commands
set -o pipefail
bash tools/traffic.sh r03_c3 --no-c4 || exit 1
bash tools/traffic.sh r03_c4 --workload c4 --steps 3 --warmup 1 || exit 2
bash tools/pmc_bench.sh r03_c3 --no-c4 || exit 3
bash tools/pmc_bench.sh r03_c4 --workload c4 --steps 3 --warmup 1 || exit 4
ls gpurun_out/r03_*traffic* gpurun_out/r03_*sq*
