#!/bin/bash
set -o pipefail
out=gpurun_out/$1; shift
mkdir -p $out
if [ "$1" = "tests" ]; then
  timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
  tail -2 $out/tests.log
fi
run() { # name, args...
  local name=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline --timing-every 1 --dump-passes "$@" > $out/$name.json 2> $out/$name.err || { tail -5 $out/$name.err; exit 2; }
}
run c3_one || exit 1
run c3_loop --emulate-shard 0/1 || exit 1
run c5big_one --workload c5big || exit 1
run c4_one --workload c4 --steps 3 --warmup 1 || exit 1
for n in 2 4 8; do
  run c4_shard${n} --workload c4 --steps 3 --warmup 1 --emulate-shard 0/$n || exit 1
done
python - $out <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    c=d['config']
    print(f.split('/')[-1], 'ms %.3f'%d['ms_per_step'], 'value %.3g'%d['value'], 'lean', c.get('lean_passes_per_registration'), c['passes']['lean_failures'])
    print('   ', c['pass_ms'][:12], 'steady', c['pass_ms'][30])
    print('   ', c['pass_kind'][:12], {k:v for k,v in c['passes'].items() if v})
PY
