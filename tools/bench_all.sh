#!/bin/bash
# ms/step of the four single-GPU configurations (no side measurements): tools/bench_all.sh [steps] [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
S=${1:-20}; shift
cd $R
COMMON="--steps $S --warmup 5 --no-cpu-baseline --no-hbm-kernels --split-engine-steps 0 --class-steps 0 --roofline-isolated-steps 0 --no-other-configs $@"
for c in "resnet101 --config $R/configs/frcnn_resnet101_coco_mtl.config" "rfcn --config $R/configs/rfcn_resnet101_voc_mtl.config" \
         "mobilenet --config $R/configs/frcnn_mobilenet_v1_voc_mtl.config" "inception --config $R/configs/frcnn_inception_resnet_v2_coco_mtl.config --height 800 --width 1333"; do
  set -- $c; name=$1; shift
  python bench.py $COMMON "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d.get('roofline',{})
print('$name', 'ms/step %.2f' % d['ms_per_step'], 'whole_step_frac %.3f' % d['whole_step']['executed_over_fp32_mfma_peak'], 'roofline_frac', round(r.get('frac',0),3), 'launch_us', round(r.get('avg_launch_us',0),1))"
done
