#!/bin/bash
# A/B of the two forward overlaps (closeness tower beside the main tower; refiner window pass on the third stream) on
# every single-GPU configuration, interleaved passes on one box.
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r05_fwd_overlap_ab.txt
echo "# ms/step, python bench.py (no side blocks); A = both forward overlaps on (rounds 3-4 default), B = MTLSSL_CLOSENESS_FWD_SIDE=0 MTLSSL_REFINE_EARLY=0" > $OUT
COMMON="--warmup 8 --no-cpu-baseline --no-hbm-kernels --split-engine-steps 0 --class-steps 0 --roofline-isolated-steps 0 --no-other-configs"
run() { tag=$1; steps=$2; cfg=$3; shift 3; env "$@" python bench.py --steps $steps $COMMON $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline',{})
print('%-28s %8.2f ms/step  in-step roofline.frac %.3f' % ('$tag', d['ms_per_step'], r.get('frac',0)))" >> $OUT; }
for pass in 1 2 3; do
  run "resnet101 A" 50 "" MTLSSL_CLOSENESS_FWD_SIDE=1 MTLSSL_REFINE_EARLY=1
  run "resnet101 B" 50 "" MTLSSL_CLOSENESS_FWD_SIDE=0 MTLSSL_REFINE_EARLY=0
done
for pass in 1 2; do
  run "rfcn A" 30 "--config configs/rfcn_resnet101_voc_mtl.config" MTLSSL_CLOSENESS_FWD_SIDE=1 MTLSSL_REFINE_EARLY=1
  run "rfcn B" 30 "--config configs/rfcn_resnet101_voc_mtl.config" MTLSSL_CLOSENESS_FWD_SIDE=0 MTLSSL_REFINE_EARLY=0
  run "rfcn closeness-side only" 30 "--config configs/rfcn_resnet101_voc_mtl.config" MTLSSL_CLOSENESS_FWD_SIDE=1 MTLSSL_REFINE_EARLY=0
  run "inception A" 20 "--config configs/frcnn_inception_resnet_v2_coco_mtl.config --height 800 --width 1333" MTLSSL_CLOSENESS_FWD_SIDE=1 MTLSSL_REFINE_EARLY=1
  run "inception B" 20 "--config configs/frcnn_inception_resnet_v2_coco_mtl.config --height 800 --width 1333" MTLSSL_CLOSENESS_FWD_SIDE=0 MTLSSL_REFINE_EARLY=0
  run "mobilenet A" 60 "--config configs/frcnn_mobilenet_v1_voc_mtl.config" MTLSSL_CLOSENESS_FWD_SIDE=1 MTLSSL_REFINE_EARLY=1
  run "mobilenet B" 60 "--config configs/frcnn_mobilenet_v1_voc_mtl.config" MTLSSL_CLOSENESS_FWD_SIDE=0 MTLSSL_REFINE_EARLY=0
done
cat $OUT
