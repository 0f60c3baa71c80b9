#!/bin/bash
# one bench per call of run(); stops at the first failure and says which
set -o pipefail
out=gpurun_out/$1; shift
mkdir -p $out
run() { # name, args...
  local name=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline --timing-every 1 --dump-passes "$@" > $out/$name.json 2> $out/$name.err || { echo "FAILED: $name"; tail -5 $out/$name.err; exit 2; }
  python - $out/$name.json <<'PY'
import json,sys
f=sys.argv[1]
d=json.loads(open(f).read().strip().splitlines()[-1])
c=d['config']
print(f.split('/')[-1], 'ms %.3f'%d['ms_per_step'], 'value %.3g'%d['value'], 'lean', c.get('lean_passes_per_registration'), c['passes']['lean_failures'], 'resid', c['residual_vs_truth'])
print('   ', c['pass_ms'][:12], 'steady', c['pass_ms'][30])
print('   ', c['pass_kind'][:12], {k:v for k,v in c['passes'].items() if v})
PY
}
