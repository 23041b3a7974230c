#!/bin/bash
# Same-box A/B of two library builds over the four single-GPU configurations: tools/ab_lib.sh <other .so> [reps] [steps]
R=${GRAFT_REPO_ROOT:-/root/repo}
OTHER=$1; REPS=${2:-2}; STEPS=${3:-20}
cd $R
for rep in $(seq $REPS); do
  echo "== this tree"; MTLSSL_AUTOTUNE=0 bash tools/bench_all.sh $STEPS 2>&1 | cut -d' ' -f1-3
  echo "== $(basename $OTHER)"; MTLSSL_LIB_PATH=$R/$OTHER MTLSSL_AUTOTUNE=0 bash tools/bench_all.sh $STEPS 2>&1 | cut -d' ' -f1-3
done
