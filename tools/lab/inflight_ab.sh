#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r05_steps_in_flight.txt
echo "# configs[1]: MTLSSL_MAX_STEPS_IN_FLIGHT (0 = unbounded, rounds 1-4) — ms/step over 50 steps; allocator after 40 steps on one batch (tools/mem_check.py)" > $OUT
for lim in 0 2 3 1 0 2; do
  MS=$(MTLSSL_MAX_STEPS_IN_FLIGHT=$lim python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-hbm-kernels --split-engine-steps 0 --class-steps 0 --roofline-isolated-steps 0 --no-other-configs 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.readline())['ms_per_step'])")
  MEM=$(MTLSSL_MAX_STEPS_IN_FLIGHT=$lim python tools/mem_check.py 40 2>&1 | grep "^allocated" )
  echo "in flight $lim: $MS ms/step; $MEM" >> $OUT
done
for cfg in "--config configs/frcnn_mobilenet_v1_voc_mtl.config" "--config configs/rfcn_resnet101_voc_mtl.config"; do
  for lim in 0 2; do
    MS=$(MTLSSL_MAX_STEPS_IN_FLIGHT=$lim python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-hbm-kernels --split-engine-steps 0 --class-steps 0 --roofline-isolated-steps 0 --no-other-configs $cfg 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "$cfg in flight $lim: $MS ms/step" >> $OUT
  done
done
cat $OUT
