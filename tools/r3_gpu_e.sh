# round 3: 8-bit asynchronous present, pipeline depth by the first launch's samples, k_gris first half at four waves, the WALK-alone probe
TAG=${1:-r3e}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "from voxel_rt2_amd import _lib; print(_lib.build_id())" > $O/build_id.txt 2>/dev/null; echo "build $(cat $O/build_id.txt)"
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py "tests/test_gpu_parity.py::test_example6_authored_grid_matches_oracle" -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit 1
run() { echo "== $*" | tee -a $O/ab.txt; env "$@" timeout -k 10 300 python tools/bench_scenes.py $CASES 2>> $O/ab.err | grep -o '"name.*"temporal_ms": [0-9.]*' | tee -a $O/ab.txt; }
for round in 1 2; do
CASES="config2_s1 scene_api"
run VRT_BENCH_STEPS=60
CASES="config2_s1"
run VRT_BENCH_STEPS=40 VRT_BENCH_FETCH_EACH=async8
run VRT_BENCH_STEPS=40 VRT_BENCH_FETCH_EACH=async
CASES="config3_s6"
run VRT_BENCH_STEPS=6
run VRT_BENCH_STEPS=6 VRT_LIB_PATH=build_variants/libvrt_grisA4.so
done
timeout -k 10 300 python tools/probes/run_walk_alone.py /tmp > $O/walk_alone.txt 2>&1; echo "walk probe rc=$?"; cat $O/walk_alone.txt
VRT_LIB_PATH=build_variants/libvrt_diag.so VRT_OVERLAP=0 timeout -k 10 300 python tools/diag_regions.py s1 dense > $O/diag_regions.txt 2>&1; echo "diag rc=$?"; cat $O/diag_regions.txt
VRT_OVERLAP=0 VRT_BENCH_STEPS=20 timeout -k 10 200 python tools/bench_scenes.py config2_s1 2>/dev/null | grep -o '"name.*"temporal_ms": [0-9.]*'
