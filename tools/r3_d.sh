#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3d
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread -k "sharded or two_ranks or rccl or no_alloc" > gpurun_out/r3d/tests.log 2>&1 || { tail -30 gpurun_out/r3d/tests.log; exit 1; }
tail -2 gpurun_out/r3d/tests.log
timeout -k 10 400 python tools/shard_probe.py --worlds 2,8 --order 2 > gpurun_out/r3d/probe2.txt 2>&1 || exit 2
cat gpurun_out/r3d/probe2.txt
run() { # name, args...
  local name=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload c4 --steps 3 --warmup 1 --timing-every 1 --dump-passes "$@" > gpurun_out/r3d/$name.json 2> gpurun_out/r3d/$name.err || { tail -5 gpurun_out/r3d/$name.err; exit 2; }
}
for n in 2 4 8; do
  run c4_shard${n}_chunks --emulate-shard 0/$n || exit 1
done
run c4_shard8_chunks_large --emulate-shard 0/8 --certify-min 1000000 || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3d/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    c=d['config']
    print(f.split('/')[-1], 'ms %.3f'%d['ms_per_step'], 'value %.3g'%d['value'], 'lean', c.get('lean_passes_per_registration'), c['passes']['lean_failures'])
    print('   ', c['pass_ms'][:12], 'steady', c['pass_ms'][30])
    print('   ', c['pass_kind'][:12], {k:v for k,v in c['passes'].items() if v})
PY
