#!/bin/bash
# the round's evidence with the final library: kernel summaries, counters, the job of N ranks in turn, the default line
set -o pipefail
bash tools/r3_prof.sh > gpurun_out/r3_prof.log 2>&1 || { echo "prof failed"; tail -5 gpurun_out/r3_prof.log; exit 1; }
bash tools/r3_prof2.sh > gpurun_out/r3_prof2.log 2>&1 || { echo "prof2 failed"; tail -5 gpurun_out/r3_prof2.log; exit 2; }
for n in 8 4 2 1; do
  timeout -k 10 400 python tools/job_probe.py --workload c4 --ranks $n > gpurun_out/r03_c4_job$n.json 2> gpurun_out/r03_c4_job$n.err || { echo "job $n failed"; exit 3; }
done
timeout -k 10 400 python tools/job_probe.py --workload c3 --ranks 8 > gpurun_out/r03_c3_job8.json 2> gpurun_out/r03_c3_job8.err || { echo "c3 job failed"; exit 4; }
timeout -k 10 600 python bench.py > gpurun_out/r03_default_bench.json 2> gpurun_out/r03_default_bench.err || { echo "bench failed"; exit 5; }
echo all done
