#!/bin/bash
# rocprofv3 kernel trace of a short default bench; leaves the sqlite db + summaries under gpurun_out/prof
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$R/gpurun_out/prof
rm -rf $P; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $P/raw -o ev -- python $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --split-engine-steps 0 "$@" > $P/bench.json 2> $P/bench.err
DB=$(find $P/raw -name "*.db" | head -1)
cp $DB $P/trace.db
cd $R
python tools/rocprof_summary.py $P/trace.db 45 > $P/kernel_stats.md
python tools/stream_gaps.py $P/trace.db 0.4 > $P/stream_gaps.txt
rm -rf $P/raw
head -c 400 $P/bench.json; echo; head -30 $P/kernel_stats.md; cat $P/stream_gaps.txt; ls -la $P
