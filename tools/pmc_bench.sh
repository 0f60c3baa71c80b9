#!/bin/bash
# Development aid: SQ counters per kernel of a bench.py command (one rocprofv3 --pmc pass, eight SQ counters).
# usage: tools/pmc_bench.sh NAME [bench.py arguments]   -> gpurun_out/NAME_sq.txt (mean per launch, by kernel)
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 280 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VALU_FMA_F64 --output-format csv -d $out/pmcb_$name -o sq -- python3 $root/bench.py --no-cpu-baseline --no-kernel-timing "$@" > $out/${name}_sq.log 2>&1 || exit 1
f=$(find $out/pmcb_$name -name "*counter_collection.csv" | head -1)
python3 - "$f" > $out/${name}_sq.txt <<'P'
import csv, re, sys
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("icpx::", "")
    d = acc.setdefault(k, {})
    v = d.setdefault(r["Counter_Name"], [0.0, 0])
    v[0] += float(r["Counter_Value"]); v[1] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", [0, 1])[0])
for k, d in rows[:10]:
    n = max(v[1] for v in d.values())
    print(k, "launches=%d" % n, " ".join("%s=%.4g" % (c, v[0] / v[1]) for c, v in sorted(d.items())))
P
rc=$?
rm -rf $out/pmcb_$name
exit $rc
