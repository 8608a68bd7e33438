#!/bin/bash
# PMC passes over one bench case (separate runs; --pmc only with --kernel-trace, per the microarch guide).
# usage: tools/pmc.sh <case-substring> <outdir-name>
set -u
CASE=${1:-config2}; TAG=${2:-pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
# counters are per kernel: isolated launches, so that a launch's counters are its own (the default schedule also completes under
# --pmc since round 2: the library leaves the dispatch gate out where its self-test finds queue operations serialised)
export VRT_OVERLAP=0
run() { name=$1; shift; timeout -k 10 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $ROOT/tools/bench_scenes.py $CASE > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU
run sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
ls $OUT/*/ | head -40
