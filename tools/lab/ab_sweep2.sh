#!/bin/bash
# Second sweep, relative to the round-5 default (forward chains serial): which of the remaining overlaps still pay.
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r05_schedule_switches_after_fwd_serial.txt
echo "# configs[1], python bench.py --steps 30 --warmup 8 (no side blocks), ms/step; default = forward chains serial; two interleaved passes, one box" > $OUT
run() { name=$1; shift; env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-hbm-kernels --split-engine-steps 0 --class-steps 0 --roofline-isolated-steps 0 --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline',{})
print('%-34s %7.2f ms/step   in-step roofline.frac %.3f' % ('$name', d['ms_per_step'], r.get('frac',0)))" >> $OUT || echo "$name failed" >> $OUT; }
for pass in 1 2; do
  echo "## pass $pass" >> $OUT
  run default X=1
  run SPLIT_LOSS=0 MTLSSL_SPLIT_LOSS=0
  run TOWER_WGRAD_STREAM=0 MTLSSL_TOWER_WGRAD_STREAM=0
  run AUX_RELEASE=late MTLSSL_AUX_RELEASE=late
  run WGRAD_STREAM=0 MTLSSL_WGRAD_STREAM=0
  run AUX_STREAM=0 MTLSSL_AUX_STREAM=0
  run KEEP_INPUT_XF=0 MTLSSL_KEEP_INPUT_XF=0
  run PLAN_DB=0 MTLSSL_PLAN_DB=0
  run REFINE_EARLY=1 MTLSSL_REFINE_EARLY=1
  run CLOSENESS_FWD_SIDE=1 MTLSSL_CLOSENESS_FWD_SIDE=1
done
cat $OUT
