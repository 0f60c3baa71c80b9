source tools/r3_h.sh r3h2
run c3_one --no-c4
run c5big_one --workload c5big
run c4_one --workload c4 --steps 3 --warmup 1
run c4_shard8 --workload c4 --steps 3 --warmup 1 --emulate-shard 0/8
python - <<'PY'
import json
for n in ("c3_one","c4_one"):
    d=json.loads(open("gpurun_out/r3h2/%s.json"%n).read().strip().splitlines()[-1])
    print(n, "build s", d["config"]["target_build_seconds"])
PY
