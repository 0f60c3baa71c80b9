#!/bin/bash
# round-3 profiles: rocprofv3 kernel summaries of the bench commands (one call each)
set -o pipefail
for n in 2 4 8; do
  bash tools/prof.sh r03_c4_shard${n} --workload c4 --steps 3 --warmup 1 --emulate-shard 0/$n --no-cpu-baseline || exit 1
done
bash tools/prof.sh r03_c3_weak_loop --emulate-shard 0/1 --no-cpu-baseline || exit 2
bash tools/prof.sh r03_c3 --no-c4 --no-cpu-baseline || exit 3
bash tools/prof.sh r03_c4 --workload c4 --steps 5 --warmup 1 --no-cpu-baseline || exit 4
bash tools/prof.sh r03_c5big --workload c5big --no-cpu-baseline || exit 5
bash tools/prof.sh r03_c6 --workload c6 --no-cpu-baseline || exit 6
ls gpurun_out/r03_*
