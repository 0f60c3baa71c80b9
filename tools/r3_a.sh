#!/bin/bash
# round-3 measurement script A: GPU tests, then C3 on one rank and as one rank's share of a multi-rank loop
set -o pipefail
mkdir -p gpurun_out/r3a
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/r3a/tests.log 2>&1 || { tail -30 gpurun_out/r3a/tests.log; exit 1; }
tail -3 gpurun_out/r3a/tests.log
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r3a/c3_one.json 2> gpurun_out/r3a/c3_one.err || exit 2
timeout -k 10 200 python bench.py --no-cpu-baseline --emulate-shard 0/1 > gpurun_out/r3a/c3_weak_loop.json 2> gpurun_out/r3a/c3_weak_loop.err || exit 3
timeout -k 10 200 python bench.py --no-cpu-baseline --emulate-shard 0/1 --lean 0 > gpurun_out/r3a/c3_weak_loop_nolean.json 2> gpurun_out/r3a/c3_weak_loop_nolean.err || exit 4
for n in 2 8; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --emulate-shard 0/$n > gpurun_out/r3a/c3_strong_${n}.json 2> gpurun_out/r3a/c3_strong_${n}.err || exit 5
  timeout -k 10 200 python bench.py --no-cpu-baseline --emulate-shard 0/$n --shard-order 0 > gpurun_out/r3a/c3_strong_${n}_blocks.json 2> gpurun_out/r3a/c3_strong_${n}_blocks.err || exit 6
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3a/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    c=d['config']
    print(f.split('/')[-1], 'ms %.3f'%d['ms_per_step'], 'value %.3g'%d['value'], 'cold', d['roofline'].get('phases',{}).get('cold_pass',{}).get('ms'), 'lean', c.get('lean_passes_per_registration'), c['passes']['lean_failures'])
PY
