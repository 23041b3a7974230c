#!/bin/bash
# One-shot evidence run on the GPU box: un-profiled default bench, rocprofv3 kernel stats of the same
# command, the PMC passes, the per-kernel conv breakdown and the other shipped configurations.
# Everything lands under gpurun_out/evidence/; copy what is to be judged into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
E=$R/gpurun_out/evidence
mkdir -p $E
cd $R
python bench.py > $E/bench_default.json 2> $E/bench_default.err
python bench.py --conv-breakdown --no-cpu-baseline > $E/bench_conv_breakdown.json 2>> $E/bench_default.err
python bench.py --no-cpu-baseline --config configs/rfcn_resnet101_voc_mtl.config > $E/bench_rfcn.json 2> $E/bench_rfcn.err
python bench.py --no-cpu-baseline --config configs/frcnn_mobilenet_v1_voc_mtl.config > $E/bench_mobilenet.json 2> $E/bench_mobilenet.err
python bench.py --no-cpu-baseline --config configs/frcnn_inception_resnet_v2_coco_mtl.config --height 800 --width 1333 > $E/bench_inception.json 2> $E/bench_inception.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $E/prof -o ev -- python $R/bench.py --no-cpu-baseline > $E/bench_profiled.json 2> $E/bench_profiled.err)
python tools/rocprof_summary.py $(find $E/prof -name "*.db" | head -1) 40 > $E/kernel_stats.md
bash tools/pmc_bench.sh > $E/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_bench 16 $E/pmc_traffic.json > $E/pmc_summary.md 2>> $E/pmc.log
rm -rf $E/prof
head -c 300 $E/bench_default.json; echo
for f in rfcn mobilenet inception; do python -c "import json,sys; d=json.load(open('$E/bench_$f.json')); print('$f', d['value'], d['ms_per_step'])"; done
head -12 $E/pmc_summary.md
