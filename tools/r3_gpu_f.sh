# round 3: WALK-alone probe with the occupancy set by the grid; presents with the "fewer than two launches running -> every slot" policy
TAG=${1:-r3f}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python tools/probes/run_walk_alone.py /tmp > $O/walk_alone.txt 2>&1; echo "walk probe rc=$?"; cat $O/walk_alone.txt
run() { echo "== $*" | tee -a $O/ab.txt; env "$@" timeout -k 10 300 python tools/bench_scenes.py $CASES 2>> $O/ab.err | grep -o '"name.*"temporal_ms": [0-9.]*' | tee -a $O/ab.txt; }
for round in 1 2; do
CASES="config2_s1 shard_1of8_config2 scene_api"
run VRT_BENCH_STEPS=60
CASES="config2_s1"
run VRT_BENCH_STEPS=40 VRT_BENCH_FETCH_EACH=1
run VRT_BENCH_STEPS=40 VRT_BENCH_FETCH_EACH=async8
run VRT_BENCH_STEPS=40 VRT_BENCH_FETCH_EACH=async8 VRT_BENCH_FETCH_LAG=2
run VRT_BENCH_STEPS=40 VRT_BENCH_FETCH_EACH=async VRT_BENCH_FETCH_LAG=2
run VRT_BENCH_STEPS=40 VRT_BENCH_SYNC_EACH=1
done
