#!/bin/bash
# Per-configuration rocprofv3 evidence (round 4): for each BASELINE.json single-GPU configuration a kernel trace of
# the bench's own schedule AND of the FULLY serialised schedule (every side stream off: each kernel alone on the chip,
# so its duration is its own), plus a step timeline. Usage: tools/config_evidence.sh [tag] [pmc]
#   -> gpurun_out/cfg_evidence/<tag>_<config>_{kernel_stats,kernel_stats_serialised,step_timeline}.{md,txt}
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
PMC=${2:-}
E=$R/gpurun_out/cfg_evidence
mkdir -p $E
cd $R
declare -A CFG
CFG[resnet101]="--config $R/configs/frcnn_resnet101_coco_mtl.config"
CFG[rfcn]="--config $R/configs/rfcn_resnet101_voc_mtl.config"
CFG[mobilenet]="--config $R/configs/frcnn_mobilenet_v1_voc_mtl.config"
CFG[inception]="--config $R/configs/frcnn_inception_resnet_v2_coco_mtl.config --height 800 --width 1333"
COMMON="--steps 8 --warmup 4 --no-cpu-baseline --no-hbm-kernels --split-engine-steps 0 --class-steps 0 --roofline-isolated-steps 0 --no-other-configs --no-roofline --join-steps 0"
for name in ${CFGS:-resnet101 rfcn mobilenet inception}; do
  args="${CFG[$name]}"
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $E/prof_$name -o ev -- python $R/bench.py $COMMON $args > $E/${TAG}_${name}_bench_profiled.json 2> $E/${name}.err)
  DB=$(find $E/prof_$name -name "*.db" | head -1)
  python tools/rocprof_summary.py $DB 60 > $E/${TAG}_${name}_kernel_stats.md
  python tools/step_timeline.py $DB 1 > $E/${TAG}_${name}_step_timeline.txt 2>/dev/null
  python tools/non_conv_breakdown.py $DB 10 > $E/${TAG}_${name}_non_conv_breakdown.md 2>/dev/null
  rm -rf $E/prof_$name
  (cd /tmp && export TMPDIR=/tmp && MTLSSL_AUX_STREAM=0 MTLSSL_WGRAD_STREAM=0 MTLSSL_SPLIT_LOSS=0 timeout 600 rocprofv3 --kernel-trace --stats -d $E/profs_$name -o ev -- python $R/bench.py $COMMON $args > $E/${TAG}_${name}_bench_profiled_serialised.json 2>> $E/${name}.err)
  DB=$(find $E/profs_$name -name "*.db" | head -1)
  python tools/rocprof_summary.py $DB 60 > $E/${TAG}_${name}_kernel_stats_serialised.md
  rm -rf $E/profs_$name
  if [ -n "$PMC" ]; then
    OUT=$E/pmc_$name; rm -rf $OUT; mkdir -p $OUT
    CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-hbm-kernels --split-engine-steps 0 --class-steps 0 --roofline-isolated-steps 0 --no-other-configs --join-steps 0 $args"
    (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 \
      --kernel-trace --output-format csv -d $OUT/sq -o sq -- $CMD > $OUT/sq.log 2>&1
     timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
     timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1
     timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/l2 -o l2 -- $CMD > $OUT/l2.log 2>&1)
    python tools/pmc_summary.py $OUT 30 $OUT/traffic.json > $E/${TAG}_${name}_pmc_summary.md 2> $OUT/summary.err
    rm -rf $OUT/sq $OUT/fetch $OUT/write $OUT/l2
  fi
  python - <<EOF
import json
for f in ("$E/${TAG}_${name}_bench_profiled.json", "$E/${TAG}_${name}_bench_profiled_serialised.json"):
    try:
        d = json.load(open(f)); print("$name", f.split("_bench_")[-1], round(d["ms_per_step"], 2), "ms/step (profiled)")
    except Exception as e:
        print("$name", f, "failed", e)
EOF
done
