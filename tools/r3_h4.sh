source tools/r3_h.sh r3h4
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/r3h4/tests.log 2>&1 || { tail -30 gpurun_out/r3h4/tests.log; exit 1; }
tail -2 gpurun_out/r3h4/tests.log
run c5big_one --workload c5big
run c4_one --workload c4 --steps 3 --warmup 1
run c4_shard8 --workload c4 --steps 3 --warmup 1 --emulate-shard 0/8
