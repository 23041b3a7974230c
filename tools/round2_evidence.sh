#!/bin/bash
# GPU tests for the touched kernels, HBM-kernel PMC passes, short bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_gpu_detection.py tests/test_gpu_postprocess.py -x -q 2>&1 | tail -3
bash tools/pmc_hbm.sh 2>&1 | tail -14
python tools/hbm_kernels.py > gpurun_out/hbm_kernels_plain.txt 2>&1; tail -1 gpurun_out/hbm_kernels_plain.txt | python -c "
import json,sys
for r in json.loads(sys.stdin.read()): print(r['kernel'][:40], round(r['avg_us'],1),'us', 'compulsory frac', round(r['frac_of_hbm_peak'],3), 'counter', r['counter_bytes'])"
