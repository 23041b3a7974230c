#!/bin/bash
# Round-4 evidence on the GPU box: bench lines of the four single-GPU configurations, per-configuration rocprofv3 kernel
# stats (bench schedule + fully serialised), step timelines, non-conv breakdowns and PMC summaries, the roofline
# kernel's HBM traffic, the HBM-bound kernels' counters, phase times, the CU-thief probe. Everything lands under
# gpurun_out/evidence4/; copy what is to be judged into profiles/ (tools/collect_round4.sh).
R=${GRAFT_REPO_ROOT:-/root/repo}
E=$R/gpurun_out/evidence4
rm -rf $E; mkdir -p $E
cd $R
python bench.py > $E/r04_bench_default.json 2> $E/bench_default.err
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --split-engine-steps 0 --config configs/rfcn_resnet101_voc_mtl.config > $E/r04_bench_rfcn_resnet101_600x1024_b4.json 2> $E/bench_rfcn.err
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --split-engine-steps 0 --config configs/frcnn_mobilenet_v1_voc_mtl.config > $E/r04_bench_mobilenet_600x1024.json 2> $E/bench_mobilenet.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --split-engine-steps 0 --config configs/frcnn_inception_resnet_v2_coco_mtl.config --height 800 --width 1333 > $E/r04_bench_inception_resnet_v2_1333x800.json 2> $E/bench_inception.err
python bench.py --steps 20 --warmup 5 --conv-breakdown --no-cpu-baseline --split-engine-steps 0 > $E/r04_bench_conv_breakdown.json 2>> $E/bench_default.err
MTLSSL_COMM_SELFTEST=1 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --split-engine-steps 0 > $E/r04_bench_rccl_1rank.json 2> $E/bench_rccl_1rank.err
python tools/phase_times.py > $E/r04_phase_times.txt 2>/dev/null
python tools/phase_times.py --config configs/frcnn_mobilenet_v1_voc_mtl.config --steps 30 >> $E/r04_phase_times.txt 2>/dev/null
python tools/cu_thief_probe.py 30 6000 > /dev/null 2>&1; cp gpurun_out/cu_thief_probe.txt $E/r04_cu_thief_probe.txt
python tools/bench_roi_fwd.py 2>/dev/null | grep -v amdgpu > $E/r04_roi_fwd_kernels.txt
bash tools/config_evidence.sh r04 pmc > $E/config_evidence.log 2>&1
cp gpurun_out/cfg_evidence/r04_* $E/ 2>/dev/null
cp gpurun_out/cfg_evidence/pmc_resnet101/traffic.json $E/r04_pmc_traffic.json 2>/dev/null
bash tools/pmc_hbm.sh > $E/pmc_hbm.log 2>&1
cp gpurun_out/pmc_hbm/hbm_kernels_pmc.json $E/r04_hbm_kernels_pmc.json 2>/dev/null
cp gpurun_out/pmc_hbm/summary.md $E/r04_pmc_hbm_kernels.md 2>/dev/null
MTLSSL_ROI_FWD=xcd bash tools/pmc_hbm.sh > $E/pmc_hbm_xcd.log 2>&1
cp gpurun_out/pmc_hbm/summary.md $E/r04_pmc_hbm_kernels_roi_fwd_xcd.md 2>/dev/null
for f in default rfcn_resnet101_600x1024_b4 mobilenet_600x1024 inception_resnet_v2_1333x800 rccl_1rank; do python -c "
import json
d=json.load(open('$E/r04_bench_$f.json')); r=d.get('roofline',{})
print('$f', round(d['value'],2), 'img/s', round(d['ms_per_step'],2), 'ms/step; whole-step frac', round(d['whole_step']['executed_over_fp32_mfma_peak'],3), '; roofline', r.get('bound'), round(r.get('frac',0),3), r.get('kernel','')[:50])"; done
ls $E | head -80
