#!/bin/bash
# Round check on the GPU box: GPU test suite, a short default bench, the 1-rank RCCL self-test bench.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/gpu_tests.txt
tail -5 gpurun_out/gpu_tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc $?"
MTLSSL_COMM_SELFTEST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_rccl1.json 2> gpurun_out/bench_rccl1.err; echo "bench rccl1 rc $?"
head -c 600 gpurun_out/bench.json; echo
python -c "import json; d=json.load(open('gpurun_out/bench_rccl1.json')); print(d['value'], d['ms_per_step'], d['data_parallel'])"
tail -3 gpurun_out/bench_rccl1.err
