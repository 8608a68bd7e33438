# A/B of the render pipeline depth (VRT_STREAMS=2: two launches in flight, 3: three) on one box
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 || exit 1
for round in 1 2; do for n in 2 3; do
  echo "== VRT_STREAMS=$n (round $round)"
  VRT_STREAMS=$n timeout -k 10 300 python tools/bench_scenes.py shard_1of8 shard_1of2 config2 sunlit_1080 config4 config5_dense 2>&1 | grep -o '"name.*"render_ms": [0-9.]*'
done; done
