#!/bin/bash
# leads of certified sets' winners: tests first, then the benches
. tools/r3_h.sh i
timeout -k 10 1000 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > $out/tests.log 2>&1 || { echo TESTS FAILED; tail -40 $out/tests.log; exit 3; }
tail -2 $out/tests.log
run c4 --workload c4 --steps 3 --warmup 1
run c5big --workload c5big --steps 10 --warmup 2
run c4s8 --workload c4 --steps 3 --warmup 1 --emulate-shard 0/8
run c3 --no-c4 --steps 10 --warmup 2
