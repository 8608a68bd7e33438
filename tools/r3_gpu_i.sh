TAG=${1:-r3i}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { echo "== $*" | tee -a $O/ab.txt; env "$@" timeout -k 10 300 python tools/bench_scenes.py $CASES 2>> $O/ab.err | grep -o '"name.*"temporal_ms": [0-9.]*' | tee -a $O/ab.txt; }
CASES="sunlit_1080p config4_dense config5_dense256 s6_sky_clouds_1080p_d8_norestir sponge256"
for round in 1 2; do
run VRT_BENCH_STEPS=12
run VRT_BENCH_STEPS=12 VRT_DENSE=0
run VRT_BENCH_STEPS=12 VRT_DENSE=1
done
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -5 $O/pytest_all.log
