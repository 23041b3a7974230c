#!/bin/bash
# Where do the tiny copy / fill launches of a step come from? Kernel trace of 3 steps; every copyBuffer / fillBuffer /
# elementwise-fill launch of the last step is printed with its neighbours on the same stream.
R=${GRAFT_REPO_ROOT:-/root/repo}
E=$R/gpurun_out/small
rm -rf $E; mkdir -p $E
cd $R
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $E/prof -o ev -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-hbm-kernels --split-engine-steps 0 --class-steps 0 --roofline-isolated-steps 0 --no-other-configs --no-roofline > $E/bench.json 2> $E/err.txt)
DB=$(find $E/prof -name "*.db" | head -1)
python tools/kernel_sequence.py $DB 0 100 > $E/seq.txt
python - <<PY
lines = open("$E/seq.txt").read().split("\n")[1:]
rows = [l for l in lines if l.strip()]
by_stream = {}
for i, l in enumerate(rows):
    st = l.split()[3]
    by_stream.setdefault(st, []).append(l)
out = []
for st, ls in sorted(by_stream.items()):
    for i, l in enumerate(ls):
        if any(k in l for k in ("copyBuffer", "fillBuffer", "FillFunctor", "elementwise")):
            out.append("stream %s: ...%s | %s | %s" % (st, ls[i - 1][38:90] if i else "-", l[:38] + l[38:100], ls[i + 1][38:90] if i + 1 < len(ls) else "-"))
open("$R/gpurun_out/r05_small_launches.txt", "w").write("\n".join(out) + "\n")
print(len(out), "small launches in the step")
print("\n".join(out[:80]))
PY
rm -rf $E/prof
