TAG=${1:-r3p}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { echo "== $*" | tee -a $O/ab.txt; env "$@" timeout -k 10 300 python tools/bench_scenes.py $CASES 2>> $O/ab.err | grep -o '"name.*"temporal_ms": [0-9.]*' | tee -a $O/ab.txt; }
CASES="config2_s1 sunlit_1080p config4_dense shard_1of8_config2 scene_api"
for round in 1 2; do
run VRT_BENCH_STEPS=30
for v in ck64 ck128 ck256; do run VRT_BENCH_STEPS=30 VRT_LIB_PATH=build_variants/libvrt_$v.so; done
done
VRT_LIB_PATH=build_variants/libvrt_ck128.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -x -q -m gpu -k "hdr_matches or traversal or full_frame_config2 or row_shards or contention or depths" > $O/pytest.log 2>&1; echo "pytest ck128 rc=$?"; tail -3 $O/pytest.log
