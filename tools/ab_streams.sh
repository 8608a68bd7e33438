# A/B of the render pipeline's depth on one box: VRT_STREAMS=2 VRT_GRID_DIV=1 (two launches of every workgroup slot in flight)
# against 4 streams with launches of a half / third / quarter of the slots each
cd $GRAFT_REPO_ROOT
for round in 1 2; do for v in "2 1" "4 2" "4 3" "4 4"; do
  set -- $v
  echo "== VRT_STREAMS=$1 VRT_GRID_DIV=$2 (round $round)"
  VRT_DEEP_ITEMS=1000000000 VRT_STREAMS=$1 VRT_GRID_DIV=$2 timeout -k 10 300 python tools/bench_scenes.py shard_1of8 shard_1of2 config2 sunlit_1080 config4 config5_dense 2>&1 | grep -o '"name.*"render_ms": [0-9.]*'
done; done
