TAG=${1:-r3k}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest "tests/test_gpu_parity.py::test_unwrapped_sqrt_and_reciprocal_over_all_floats" -x -q -m gpu > $O/pytest_selftest.log 2>&1; rc=$?; echo "selftest rc=$rc"; tail -4 $O/pytest_selftest.log
[ $rc -ne 0 ] && exit 1
run() { echo "== $*" | tee -a $O/ab.txt; env "$@" timeout -k 10 300 python tools/bench_scenes.py $CASES 2>> $O/ab.err | grep -o '"name.*"temporal_ms": [0-9.]*' | tee -a $O/ab.txt; }
CASES="config2_s1 sunlit_1080p config4_dense config5_dense256 config3_s6 s6_sky_clouds_1080p_d8_norestir scene_api"
for round in 1 2; do
run VRT_BENCH_STEPS=20
run VRT_BENCH_STEPS=20 VRT_LIB_PATH=build_variants/libvrt_plain.so
done
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -5 $O/pytest_all.log
