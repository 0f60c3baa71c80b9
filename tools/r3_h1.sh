source tools/r3_h.sh r3h
run c3_one
run c3_loop --emulate-shard 0/1
run c5big_one --workload c5big
run c5big_old --workload c5big --noisy-policy 0
run c4_one --workload c4 --steps 3 --warmup 1
for n in 2 4 8; do
  run c4_shard${n} --workload c4 --steps 3 --warmup 1 --emulate-shard 0/$n
done
