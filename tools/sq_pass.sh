#!/bin/bash
# SQ counter pass over the scoring kernels of one workload: tools/sq_pass.sh <tag> [workload]
T=${1:-sq}; W=${2:-config4}
O=gpurun_out/$T
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $O
FSEG_NO_FORK=1 timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${W}_sq -o p -- python3 bench.py --workload $W --contexts 1 --no-cpu-baseline --no-e2e --no-extras --steps 8 > /dev/null 2> $O/pmc_${W}_sq.err
python tools/sq_summary.py $O/pmc_${W}_sq/p_counter_collection.csv $O/pmc_${W}_sq/p_kernel_trace.csv > $O/${W}_sq_summary.txt
cat $O/${W}_sq_summary.txt
rm -rf $O/pmc_${W}_sq
