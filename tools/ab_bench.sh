#!/bin/bash
# A/B of the step-level switches on the GPU box (short benches, no CPU baseline).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err; python -c "import json; d=json.load(open('gpurun_out/ab_$name.json')); print('$name', round(d['ms_per_step'],2), 'ms/step', round(d['value'],2), 'img/s', 'roof', round(d['roofline']['achieved'],1))"; }
run base X=1
run nowgs MTLSSL_WGRAD_STREAM=0
run base2 X=1
python -m pytest tests/test_gpu_model.py tests/test_gpu_detection.py tests/test_gpu_data_parallel.py tests/test_gpu_comm.py -x -q 2>&1 | tail -3
