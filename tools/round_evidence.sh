#!/bin/bash
# One-shot evidence run on the GPU box (round 3): un-profiled default bench, rocprofv3 kernel stats + step timeline
# of the same command, the PMC passes (roofline kernel + HBM-bound kernels), the per-kernel conv breakdown, the
# 1-rank RCCL self-test bench and the other shipped configurations. Everything lands under gpurun_out/evidence/;
# copy what is to be judged into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
E=$R/gpurun_out/evidence
rm -rf $E; mkdir -p $E
cd $R
python bench.py --split-engine-steps 0 > $E/bench_default.json 2> $E/bench_default.err
python bench.py --steps 20 --warmup 5 --conv-breakdown --no-cpu-baseline > $E/bench_conv_breakdown.json 2>> $E/bench_default.err
MTLSSL_COMM_SELFTEST=1 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $E/bench_rccl_1rank.json 2> $E/bench_rccl_1rank.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --config configs/rfcn_resnet101_voc_mtl.config > $E/bench_rfcn.json 2> $E/bench_rfcn.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --config configs/frcnn_mobilenet_v1_voc_mtl.config > $E/bench_mobilenet.json 2> $E/bench_mobilenet.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --config configs/frcnn_inception_resnet_v2_coco_mtl.config --height 800 --width 1333 > $E/bench_inception.json 2> $E/bench_inception.err
python tools/phase_times.py > $E/phase_times.txt 2>/dev/null
python tools/phase_times.py --config configs/frcnn_mobilenet_v1_voc_mtl.config --steps 30 >> $E/phase_times.txt 2>/dev/null
# profile of the timed region's schedule only (no side measurements that switch overlaps off)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $E/prof -o ev -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-hbm-kernels --split-engine-steps 0 --class-steps 0 --roofline-isolated-steps 0 --no-other-configs > $E/bench_profiled.json 2> $E/bench_profiled.err)
# the same with the two forward overlaps off: every launch of the roofline kernel alone on the chip (roofline.isolated)
(cd /tmp && export TMPDIR=/tmp && MTLSSL_CLOSENESS_FWD_SIDE=0 MTLSSL_REFINE_EARLY=0 rocprofv3 --kernel-trace --stats -d $E/prof_ser -o ev -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-hbm-kernels --split-engine-steps 0 --class-steps 0 --roofline-isolated-steps 0 --no-other-configs > $E/bench_profiled_serialised.json 2> $E/bench_profiled_serialised.err)
DBS=$(find $E/prof_ser -name "*.db" | head -1)
python tools/rocprof_summary.py $DBS 45 > $E/kernel_stats_forward_serialised.md
rm -rf $E/prof_ser
DB=$(find $E/prof -name "*.db" | head -1)
python tools/rocprof_summary.py $DB 45 > $E/kernel_stats.md
python tools/step_timeline.py $DB 1 > $E/step_timeline.txt
python tools/kernel_sequence.py $DB 0 60 > $E/kernel_sequence.txt
python tools/bench_roi_bwd.py > $E/roi_bwd.txt 2>/dev/null
python tools/stream_gaps.py $DB 0.3 > $E/stream_gaps.txt
rm -rf $E/prof
bash tools/pmc_bench.sh > $E/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_bench 18 $E/pmc_traffic.json > $E/pmc_summary.md 2>> $E/pmc.log
bash tools/pmc_hbm.sh > $E/pmc_hbm.log 2>&1
cp gpurun_out/pmc_hbm/hbm_kernels_pmc.json gpurun_out/pmc_hbm/summary.md $E/ 2>/dev/null
head -c 300 $E/bench_default.json; echo
for f in rfcn mobilenet inception rccl_1rank; do python -c "import json,sys; d=json.load(open('$E/bench_$f.json')); print('$f', d['value'], d['ms_per_step'])"; done
head -12 $E/pmc_summary.md
