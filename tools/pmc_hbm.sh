#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) for the HBM-bound kernels: tools/hbm_kernels.py
# under rocprofv3, summarised to gpurun_out/pmc_hbm/hbm_kernels_pmc.json (copy to profiles/r03_hbm_kernels_pmc.json).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_hbm
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/hbm_kernels.py 4"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/l2 -o l2 -- $CMD > $OUT/l2.log 2>&1
cd $R
python tools/pmc_hbm_summary.py $OUT > $OUT/summary.md
cat $OUT/summary.md
find $OUT -name "*.csv" -size +8M -delete
