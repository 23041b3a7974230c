#!/bin/bash
# Round-6 evidence on the GPU box: the driver's own command (default bench line incl. other_configs and the join timer),
# per-configuration rocprofv3 kernel stats in both schedules (+ PMC for configs[1] and configs[2]), per-shape layer tables,
# the HBM-bound kernels' counters. Everything lands under gpurun_out/evidence6/; copy what is to be judged into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
E=$R/gpurun_out/evidence6
mkdir -p $E
cd $R
python bench.py > $E/r06_bench_default.json 2> $E/bench_default.err
CFGS="resnet101 rfcn" bash tools/config_evidence.sh r06 pmc > $E/config_evidence.log 2>&1
CFGS="mobilenet inception" bash tools/config_evidence.sh r06 >> $E/config_evidence.log 2>&1
cp gpurun_out/cfg_evidence/r06_* $E/ 2>/dev/null
cp gpurun_out/cfg_evidence/pmc_resnet101/traffic.json $E/r06_pmc_traffic.json 2>/dev/null
for c in resnet101 rfcn inception; do python tools/layer_times.py $c 4 > $E/r06_layer_times_$c.md 2>/dev/null; done
bash tools/pmc_hbm.sh > $E/pmc_hbm.log 2>&1
cp gpurun_out/pmc_hbm/hbm_kernels_pmc.json $E/r06_hbm_kernels_pmc.json 2>/dev/null
cp gpurun_out/pmc_hbm/summary.md $E/r06_pmc_hbm_kernels.md 2>/dev/null
python - <<PY
import json
d = json.load(open("$E/r06_bench_default.json")); r = d.get("roofline", {})
print("default", round(d["value"], 2), "img/s", round(d["ms_per_step"], 2), "ms/step; whole-step", round(d["whole_step"]["executed_over_fp32_mfma_peak"], 3),
      "; roofline in-step", round(r.get("frac", 0), 3), "isolated", round(r.get("frac_isolated", 0), 3), "; main-stream idle", d["whole_step"].get("main_stream_idle_ms"))
for k, v in d.get("other_configs", {}).items():
    print(" ", k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("ms_per_step", "images_per_sec", "executed_over_fp32_mfma_peak", "error")},
          "roofline", round(v.get("roofline", {}).get("frac", 0), 3), v.get("roofline", {}).get("kernel", "")[:50])
print("cpu_baseline", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
PY
ls $E
