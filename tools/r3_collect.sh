#!/bin/bash
# copies what tools/r3_final.sh left under gpurun_out/ to the names kept under profiles/
set -e
for f in r03_c3_bench_under_rocprof.json r03_c3_kernel_stats.csv r03_c3_traffic.json r03_c3_weak_loop_bench_under_rocprof.json r03_c3_weak_loop_kernel_stats.csv r03_c4_bench_under_rocprof.json r03_c4_kernel_stats.csv r03_c4_shard2_bench_under_rocprof.json r03_c4_shard2_kernel_stats.csv r03_c4_shard4_bench_under_rocprof.json r03_c4_shard4_kernel_stats.csv r03_c4_shard8_bench_under_rocprof.json r03_c4_shard8_kernel_stats.csv r03_c4_traffic.json r03_c5big_bench_under_rocprof.json r03_c5big_kernel_stats.csv r03_c6_bench_under_rocprof.json r03_c6_kernel_stats.csv r03_default_bench.json; do cp gpurun_out/$f profiles/$f; done
for n in 1 2 4 8; do cp gpurun_out/r03_c4_job$n.json profiles/r03_c4_job${n}_one_gpu_in_turn.json; done
cp gpurun_out/r03_c3_job8.json profiles/r03_c3_job8_one_gpu_in_turn.json
cp gpurun_out/r03_c3_sq.txt profiles/r03_pmc_c3_bench_sq_by_kernel.txt
cp gpurun_out/r03_c4_sq.txt profiles/r03_pmc_c4_bench_sq_by_kernel.txt
