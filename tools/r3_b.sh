#!/bin/bash
# round-3 measurement script B: C4 on one rank and as one rank's share of 2, 4, 8
set -o pipefail
mkdir -p gpurun_out/r3b
run() { # name, args...
  local name=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload c4 --steps 3 --warmup 1 --timing-every 1 --dump-passes "$@" > gpurun_out/r3b/$name.json 2> gpurun_out/r3b/$name.err || { tail -5 gpurun_out/r3b/$name.err; exit 2; }
  echo "$name done"
}
run c4_one || exit 1
for n in 2 4 8; do
  run c4_shard${n}_curve --emulate-shard 0/$n || exit 1
  run c4_shard${n}_blocks --emulate-shard 0/$n --shard-order 0 || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3b/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    c=d['config']
    print(f.split('/')[-1], 'ms %.3f'%d['ms_per_step'], 'value %.3g'%d['value'], 'lean', c.get('lean_passes_per_registration'), c['passes']['lean_failures'])
    print('   ', c['pass_ms'][:12], 'steady', c['pass_ms'][30])
    print('   ', c['pass_kind'][:12], {k:v for k,v in c['passes'].items() if v})
PY
