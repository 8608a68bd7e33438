#!/bin/bash
# round 3, resampling loop of k_gris<.,.,2> running through its dead taps first: ReSTIR parity, then the three ReSTIR scenes twice
TAG=${1:-r3q}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "restir or reference or golden or config3" > $O/pytest_restir.log 2>&1; echo "pytest restir rc=$?"; tail -3 $O/pytest_restir.log
run() { echo "== $*" | tee -a $O/ab.txt; env "$@" timeout -k 10 300 python tools/bench_scenes.py $CASES 2>> $O/ab.err | grep -o '"name.*"temporal_ms": [0-9.]*' | tee -a $O/ab.txt; }
CASES="config3_s6 sunlit_restir s6_nosky_restir"
for round in 1 2; do
run VRT_BENCH_STEPS=8
done
