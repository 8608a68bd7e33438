TAG=${1:-r3n}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { echo "== $*" | tee -a $O/ab.txt; env "$@" timeout -k 10 300 python tools/bench_scenes.py $CASES 2>> $O/ab.err | grep -o '"name.*"temporal_ms": [0-9.]*' | tee -a $O/ab.txt; }
CASES="config2_s1 sunlit_1080p config4_dense s6_sky_clouds_1080p_d8_norestir"
for round in 1 2; do
run VRT_BENCH_STEPS=20
for v in g128 g192 g256; do run VRT_BENCH_STEPS=20 VRT_LIB_PATH=build_variants/libvrt_$v.so; done
done
VRT_LIB_PATH=build_variants/libvrt_g256.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hdr_matches or traversal or full_frame_config2" > $O/pytest.log 2>&1; echo "pytest g256 rc=$?"; tail -3 $O/pytest.log
