# A/B of a build variant against the shipped library: parity first (with a timeout), then the bench scenes
V=$1; shift
cd $GRAFT_REPO_ROOT
echo "== parity with $V"; VRT_LIB_PATH=build_variants/libvrt_$V.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grid256.py -x -q -k "hdr_matches or full_frame_config2 or counters or row_shards or config5" 2>&1 | tail -3
for v in shipped $V; do
  echo "== $v"
  if [ $v == shipped ]; then unset VRT_LIB_PATH; else export VRT_LIB_PATH=build_variants/libvrt_$v.so; fi
  timeout -k 10 300 python tools/bench_scenes.py config2 sunlit_1080 config4 config5_dense 2>&1 | grep -o '"name.*"render_ms": [0-9.]*'
done
