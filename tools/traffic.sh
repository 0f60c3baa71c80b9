#!/bin/bash
# Development aid: memory-side bytes per kernel of a bench.py command, one rocprofv3 --pmc
# pass for FETCH_SIZE and one for WRITE_SIZE.
# usage: tools/traffic.sh NAME [bench.py arguments]   -> gpurun_out/NAME_traffic.json
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 280 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${name}_$c -o $c -- python3 $root/bench.py --no-cpu-baseline --no-kernel-timing "$@" > $out/${name}_$c.log 2>&1 || exit 1
done
f=$(find $out/pmc_${name}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find $out/pmc_${name}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 $root/tools/traffic_kernels.py $f $w $out/${name}_traffic.json
rc=$?
rm -rf $out/pmc_${name}_FETCH_SIZE $out/pmc_${name}_WRITE_SIZE
exit $rc
