#!/bin/bash
# Two SQ counter passes (lane utilisation, instruction mix) over one bench case: tools/pmc_sq.sh <case-substring> <outdir-name>
set -u
CASE=${1:-config2}; TAG=${2:-pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export VRT_OVERLAP=0
run() { name=$1; shift; timeout -k 10 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $ROOT/tools/bench_scenes.py $CASE > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU && \
run sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS
python $ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt
