TAG=${1:-r3j}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { echo "== $*" | tee -a $O/ab.txt; env "$@" timeout -k 10 300 python tools/bench_scenes.py $CASES 2>> $O/ab.err | grep -o '"name.*"temporal_ms": [0-9.]*' | tee -a $O/ab.txt; }
CASES="config4_dense config5_dense256"
for round in 1 2; do
run VRT_BENCH_STEPS=12
run VRT_BENCH_STEPS=12 VRT_LIB_PATH=build_variants/libvrt_walkbr.so
done
