#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python command: tools/prof_cmd.sh <name> <python args...>
# -> gpurun_out/prof_<name>/kernel_stats.md
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
P=$R/gpurun_out/prof_$NAME
rm -rf $P; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $P/raw -o ev -- python "$@" > $P/stdout.txt 2> $P/stderr.txt
DB=$(find $P/raw -name "*.db" | head -1)
cp $DB $P/trace.db
cd $R
python tools/rocprof_summary.py $P/trace.db 40 > $P/kernel_stats.md
rm -rf $P/raw
head -20 $P/kernel_stats.md
