#!/bin/bash
# Runs every lab binary under tools/lab/bin on the GPU box; output -> gpurun_out/lab.log
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for b in $(ls tools/lab/bin | sort -t_ -k2 -n); do
  echo "=== $b"; timeout 300 tools/lab/bin/$b
done > gpurun_out/lab.log 2>&1
tail -5 gpurun_out/lab.log
