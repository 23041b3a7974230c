#!/bin/bash
# Same-box A/B of environment switches on one configuration: tools/ab_env.sh <config> <steps> <reps> "NAME:VAR=V,VAR2=V2" ...
#   config: resnet101 | rfcn | mobilenet | inception ; the spec "default:" runs with no extra variable
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; STEPS=$2; REPS=$3; shift 3
declare -A CFG
CFG[resnet101]="configs/frcnn_resnet101_coco_mtl.config"
CFG[rfcn]="configs/rfcn_resnet101_voc_mtl.config"
CFG[mobilenet]="configs/frcnn_mobilenet_v1_voc_mtl.config"
CFG[inception]="configs/frcnn_inception_resnet_v2_coco_mtl.config --height 800 --width 1333"
COMMON="--steps $STEPS --warmup 6 --no-cpu-baseline --no-hbm-kernels --split-engine-steps 0 --class-steps 0 --roofline-isolated-steps 0 --no-other-configs --no-roofline --join-steps 0"
cd $R
for rep in $(seq $REPS); do
  for spec in "$@"; do
    tag=${spec%%:*}; envs=${spec#*:}; envs=${envs//,/ }
    env $envs python bench.py $COMMON --config ${CFG[$NAME]} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-10s %-34s %8.2f ms/step' % ('$NAME', '$tag', d['ms_per_step']))" || echo "$NAME $tag failed"
  done
done
