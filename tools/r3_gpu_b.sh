# round 3: the carried schedule -- pipeline tests first, then the A/B against the overlapped pipeline, then the whole GPU suite
TAG=${1:-r3b}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "from voxel_rt2_amd import _lib; print(_lib.build_id())" > $O/build_id.txt 2>/dev/null; echo "build $(cat $O/build_id.txt)"
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu > $O/pytest_pipeline.log 2>&1; rc=$?; echo "pytest pipeline rc=$rc"; tail -5 $O/pytest_pipeline.log
[ $rc -ne 0 ] && exit 1
for round in 1 2; do for pipe in overlap carry; do
  echo "== VRT_PIPE=$pipe (round $round)"
  VRT_PIPE=$pipe VRT_BENCH_STEPS=40 timeout -k 10 300 python tools/bench_scenes.py config2_s1 shard_1of8 shard_1of2 sunlit_1080p config4_dense 2>> $O/ab.err | grep -o '"name.*"temporal_ms": [0-9.]*' | tee -a $O/ab_$pipe.txt
done; done
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -5 $O/pytest_all.log
