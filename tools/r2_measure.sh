# round-2 measurement pass: bench line, kernel trace of the same command, counter passes per config, scene table.
# Everything lands in gpurun_out/$1 (default r2_final); tools/make_traffic.py + a copy into profiles/ follow on the build host.
TAG=${1:-r2_final}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "from voxel_rt2_amd import _lib; print(_lib.build_id())" > $O/build_id.txt 2>/dev/null
echo "build $(cat $O/build_id.txt)"
for c in config2_s1 config5_dense256 config4_dense config3_s6; do bash tools/pmc.sh $c ${TAG}_pmc_$c > /dev/null 2>&1; python tools/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$c > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$c/summary.txt; echo "pmc $c: $(grep -c mean $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$c/summary.txt) rows"; done
python tools/bench_scenes.py > $O/scenes.jsonl 2>/dev/null; echo "scenes: $(grep -c name $O/scenes.jsonl)"
VRT_BENCH_RESERVE=8 python tools/bench_scenes.py config2_s1 2>/dev/null | sed 's/config2_s1_1080p_d8/config2_s1_1080p_d8_reserve8/' >> $O/scenes.jsonl
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> $O/trace.err; echo "trace rc=$?"
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU --output-format csv -d $O/pmc_default -o p -- python $GRAFT_REPO_ROOT/tools/probe_overlap.py > $O/pmc_default_schedule.txt 2>&1; echo "pmc default schedule rc=$?"
