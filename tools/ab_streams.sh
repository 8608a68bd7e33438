# A/B of the render pipeline's depth on one box: VRT_STREAMS=2 VRT_GRID_DIV=1 (two launches of every workgroup slot in flight)
# against 4 streams with launches of half the slots each.  usage: tools/ab_streams.sh [steps] -- case...
cd $GRAFT_REPO_ROOT
steps=${1:-16}; shift; [ "$1" == "--" ] && shift
cases=("$@"); [ ${#cases[@]} -eq 0 ] && cases=(shard_1of8 shard_1of2 config2 sunlit_1080 s6_nosky_plain config4)
for round in 1 2; do for v in "2 1" "4 2"; do
  set -- $v
  echo "== VRT_STREAMS=$1 VRT_GRID_DIV=$2, $steps steps (round $round)"
  VRT_BENCH_STEPS=$steps VRT_STREAMS=$1 VRT_GRID_DIV=$2 timeout -k 10 400 python tools/bench_scenes.py "${cases[@]}" 2>&1 | grep -o '"name.*"render_ms": [0-9.]*'
done; done
