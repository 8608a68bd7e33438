# round 3: one-sample launches pipelined, eight-launch pipeline on small frames, asynchronous presents, tiles written by the temporal pass
TAG=${1:-r3d}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "from voxel_rt2_amd import _lib; print(_lib.build_id())" > $O/build_id.txt 2>/dev/null; echo "build $(cat $O/build_id.txt)"
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py tests/test_bench_multi.py -x -q -m gpu > $O/pytest_pipeline.log 2>&1; rc=$?; echo "pytest pipeline rc=$rc"; tail -5 $O/pytest_pipeline.log
[ $rc -ne 0 ] && exit 1
run() { echo "== $*" | tee -a $O/ab.txt; env "$@" timeout -k 10 300 python tools/bench_scenes.py $CASES 2>> $O/ab.err | grep -o '"name.*"temporal_ms": [0-9.]*' | tee -a $O/ab.txt; }
for round in 1 2; do
CASES="config2_s1 shard_1of8 shard_1of2 scene_api"
run VRT_BENCH_STEPS=40
run VRT_BENCH_STEPS=40 VRT_DEEPER_ITEMS=0
run VRT_BENCH_STEPS=40 VRT_OVERLAP_SINGLE=0
CASES="config2_s1 config4_dense"
run VRT_BENCH_STEPS=30 VRT_BENCH_FETCH_EACH=1
run VRT_BENCH_STEPS=30 VRT_BENCH_FETCH_EACH=async
done
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_pipeline.py --deselect tests/test_bench_multi.py > $O/pytest_all.log 2>&1; echo "pytest rest rc=$?"; tail -5 $O/pytest_all.log
