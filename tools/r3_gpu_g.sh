TAG=${1:-r3g}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { echo "== $*" | tee -a $O/ab.txt; env "$@" timeout -k 10 300 python tools/bench_scenes.py $CASES 2>> $O/ab.err | grep -o '"name.*"temporal_ms": [0-9.]*' | tee -a $O/ab.txt; }
CASES="config2_s1"
for round in 1 2; do for fb in 1 2 3; do for lag in 1 2 3; do
run VRT_BENCH_STEPS=60 VRT_BENCH_FETCH_EACH=async8 VRT_BENCH_FETCH_LAG=$lag VRT_FULL_BELOW=$fb
done; done; done
