TAG=${1:-r3m}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { echo "== $*" | tee -a $O/ab.txt; env "$@" timeout -k 10 300 python tools/bench_scenes.py $CASES 2>> $O/ab.err | grep -o '"name.*"temporal_ms": [0-9.]*' | tee -a $O/ab.txt; }
CASES="config3_s6 sunlit_restir s6_nosky_restir"
for round in 1 2; do
run VRT_BENCH_STEPS=8
done
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -5 $O/pytest_all.log
