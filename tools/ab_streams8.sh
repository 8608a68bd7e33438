# eight render streams against four (needs a library built with -DVRT_MAX_STREAMS=8 and VRT_STREAMS accepting 8 in vrt_api.hip's
# ensure_overlap: build_variants/libvrt_s8.so, not the shipped build); numbers in profiles/r02_pipeline_depth.txt
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
for round in 1 2; do for v in "4 2" "8 4" "8 3" "8 2"; do
  set -- $v
  echo "== VRT_STREAMS=$1 VRT_GRID_DIV=$2 (round $round)"
  VRT_LIB_PATH=build_variants/libvrt_s8.so VRT_BENCH_STEPS=40 VRT_STREAMS=$1 VRT_GRID_DIV=$2 timeout -k 10 300 python tools/bench_scenes.py shard_1of8 shard_1of2 config2 2>&1 | grep -o '"name.*"render_ms": [0-9.]*'
done; done
