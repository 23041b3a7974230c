#!/bin/bash
# PMC counters for the conv microbench (separate pass from kernel-trace, per gpurun rules)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|GRBM_GUI_ACTIVE|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY|SQ_WAIT_ANY|SQ_ACTIVE_INST_ANY|LDS_BANK|FETCH_SIZE|WRITE_SIZE" | head -40 > gpurun_out/pmc_list.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc1 -o pmc -- python tools/bench_conv.py --iters 3 --only "2560" > gpurun_out/pmc1.log 2>&1
ls -R gpurun_out/pmc1 | head
