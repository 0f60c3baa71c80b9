#!/bin/bash
. tools/r3_h.sh j
run c4 --workload c4 --steps 3 --warmup 1
run c5big --workload c5big --steps 10 --warmup 2
run c4s8 --workload c4 --steps 3 --warmup 1 --emulate-shard 0/8
run c3 --no-c4 --steps 10 --warmup 2
