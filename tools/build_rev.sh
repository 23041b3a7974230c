#!/bin/bash
# Build the library of another git revision for same-box A/B runs: tools/build_rev.sh <rev> -> tools/lab/bin/libmtlssl_hip_<rev>.so
# (bench with MTLSSL_LIB_PATH=<that file>; the C ABI has to be the same).
set -e
R=$(cd $(dirname $0)/.. && pwd)
T=/tmp/mtlssl_rev_$1
rm -rf $T && mkdir -p $T
(cd $R && git archive $1) | tar -x -C $T
(cd $T && python -c "
import sys; sys.path.insert(0, '.')
from mtl_ssl_amd import build; build.build(verbose=False)")
mkdir -p $R/tools/lab/bin
cp $T/mtl_ssl_amd/libmtlssl_hip.so $R/tools/lab/bin/libmtlssl_hip_$1.so
echo $R/tools/lab/bin/libmtlssl_hip_$1.so
