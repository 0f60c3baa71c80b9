#!/bin/bash
# Development aid: rocprofv3 kernel summary of a bench.py command.
# usage: tools/prof.sh NAME [bench.py arguments]   -> gpurun_out/NAME_kernel_stats.csv, NAME_bench_under_rocprof.json
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -o $name -- python3 $root/bench.py "$@" > $out/${name}_bench_under_rocprof.json 2> $out/${name}_bench_under_rocprof.err
rc=$?
f=$(find $out/prof_$name -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $out/${name}_kernel_stats.csv
rm -rf $out/prof_$name
exit $rc
