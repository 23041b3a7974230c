#!/bin/bash
# Same-box A/B of THIS tree against another checkout of the repository under ab_base/ (git archive <rev> | tar -x -C ab_base,
# built there; ab_base/ is git-ignored but travels with gpurun): tools/ab_tree.sh <config name> [reps] [steps] [extra bench args]
#   config: resnet101 | rfcn | mobilenet | inception
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=${1:-resnet101}; REPS=${2:-2}; STEPS=${3:-20}; shift 3
declare -A CFG
CFG[resnet101]="configs/frcnn_resnet101_coco_mtl.config"
CFG[rfcn]="configs/rfcn_resnet101_voc_mtl.config"
CFG[mobilenet]="configs/frcnn_mobilenet_v1_voc_mtl.config"
CFG[inception]="configs/frcnn_inception_resnet_v2_coco_mtl.config --height 800 --width 1333"
COMMON="--steps $STEPS --warmup 6 --no-cpu-baseline --no-hbm-kernels --split-engine-steps 0 --class-steps 0 --roofline-isolated-steps 0 --no-other-configs --no-roofline"
one() { (cd $1 && python bench.py $COMMON --config ${CFG[$NAME]} "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-10s %-9s %8.2f ms/step' % ('$NAME', '$(basename $1)', d['ms_per_step']))"); }
for rep in $(seq $REPS); do
  one $R "$@"
  one $R/ab_base "$@"
done
