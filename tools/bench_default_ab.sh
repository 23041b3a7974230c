#!/bin/bash
# The driver's own command (python bench.py, defaults) on this tree and on round 5's tree (ab_base/) on the SAME box:
# boxes of the pool differ by a few per cent, so a round-over-round comparison of the headline needs both on one box.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/bench_ab
mkdir -p $O
cd $R && python bench.py > $O/r06_bench_default.json 2> $O/this.err
cd $R/ab_base && python bench.py --no-other-configs > $O/r06_bench_default_round5_tree_same_box.json 2> $O/base.err
cd $R && python bench.py --no-other-configs --no-cpu-baseline > $O/r06_bench_default_second_run.json 2>> $O/this.err
python - <<PY
import json
for f in ("r06_bench_default.json", "r06_bench_default_round5_tree_same_box.json", "r06_bench_default_second_run.json"):
    d = json.load(open("$O/" + f)); r = d.get("roofline", {})
    print("%-52s %6.2f ms/step %6.2f img/s  roofline kernel %.0f us in-step (frac %.3f)" % (f, d["ms_per_step"], d["value"], r.get("avg_launch_us", 0), r.get("frac", 0)))
PY
