#!/bin/bash
# A/B of the step-level switches on the GPU box (short benches, no CPU baseline).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err; python -c "import json; d=json.load(open('gpurun_out/ab_$name.json')); print('$name', round(d['ms_per_step'],2), 'ms/step', round(d['value'],2), 'img/s', 'roof', round(d['roofline']['achieved'],1))" || tail -5 gpurun_out/ab_$name.err; }
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}; envs=${envs//,/ }
  run $name $envs
done
