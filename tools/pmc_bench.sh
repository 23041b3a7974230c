#!/bin/bash
# PMC passes over the headline bench (each counter group in its own rocprofv3 run, --kernel-trace only,
# as MI355X_MICROARCH.md's rocprofv3 section prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_bench
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --split-engine-steps 0 --join-steps 0"
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 \
  --kernel-trace --output-format csv -d $OUT/sq -o sq -- $CMD > $OUT/sq.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/l2 -o l2 -- $CMD > $OUT/l2.log 2>&1
ls -R $OUT | head -40
for f in sq fetch write l2; do tail -2 $OUT/$f.log; done
find $OUT -name "*.csv" -size +20M -delete
