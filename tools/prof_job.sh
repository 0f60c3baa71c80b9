#!/bin/bash
# Development aid: rocprofv3 kernel summary of tools/job_probe.py.  usage: tools/prof_job.sh NAME [job_probe arguments]
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -o $name -- python3 $root/tools/job_probe.py "$@" > $out/${name}_job.json 2> $out/${name}_job.err
rc=$?
f=$(find $out/prof_$name -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $out/${name}_kernel_stats.csv
rm -rf $out/prof_$name
exit $rc
