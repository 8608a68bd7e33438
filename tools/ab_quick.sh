# A/B of build variants against the shipped library on the bench scenes (no parity run): tools/ab_quick.sh v1 [v2 ...] -- case...
names=(shipped); cases=(config2 sunlit_1080 config4 config5_dense)
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; cases=("$@"); break; fi; names+=("$1"); shift; done
cd $GRAFT_REPO_ROOT
for round in 1 2; do for v in "${names[@]}"; do
  if [ $v == shipped ]; then unset VRT_LIB_PATH; else export VRT_LIB_PATH=build_variants/libvrt_$v.so; fi
  echo "== $v (round $round)"; timeout -k 10 300 python tools/bench_scenes.py "${cases[@]}" 2>&1 | grep -o '"name.*"render_ms": [0-9.]*' | grep -v shard
done; done
