#!/bin/bash
# certified sets weighed by the finalize (matrix-core build) / pairs screened in k_contested
. tools/r3_h.sh i
timeout -k 10 900 python -u -m pytest tests/test_lean.py tests/test_shards.py tests/test_gpu_parity.py -x -q -m gpu --timeout 300 --timeout-method thread -k "lean or shard or policy or certif or device_loop or full_size or finalize" > $out/tests.log 2>&1 || { echo TESTS FAILED; tail -30 $out/tests.log; exit 3; }
tail -2 $out/tests.log
run c4 --workload c4 --steps 3 --warmup 1
run c5big --workload c5big --steps 10 --warmup 2
run c4s8 --workload c4 --steps 3 --warmup 1 --emulate-shard 0/8
run c4s4 --workload c4 --steps 3 --warmup 1 --emulate-shard 0/4
