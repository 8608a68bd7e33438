TAG=${1:-r3h}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { echo "== $*" | tee -a $O/ab.txt; env "$@" timeout -k 10 300 python tools/bench_scenes.py $CASES 2>> $O/ab.err | grep -o '"name.*"temporal_ms": [0-9.]*' | tee -a $O/ab.txt; }
CASES="sunlit_1080p config4_dense config5_dense256 s6_sky_clouds_1080p_d8_norestir config3_s6"
for round in 1 2; do
run VRT_BENCH_STEPS=12
run VRT_BENCH_STEPS=12 VRT_LIB_PATH=build_variants/libvrt_shbr.so
done
VRT_LIB_PATH=build_variants/libvrt_shbr.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hdr_matches or restir or traversal" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
