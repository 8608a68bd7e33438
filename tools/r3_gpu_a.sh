# round 3, first GPU pass (shipped round-2 kernels): full-size sky tables against the oracle, kernel traces of the secondary
# configs (profiles/r03_*_config{3,4,5}_kernel_stats.csv), shard baselines for the pipeline work.
TAG=${1:-r3a}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "from voxel_rt2_amd import _lib; print(_lib.build_id())" > $O/build_id.txt 2>/dev/null; echo "build $(cat $O/build_id.txt)"
VRT_BENCH_STEPS=40 python tools/bench_scenes.py config2_s1 shard_1of8 shard_1of2 > $O/scenes_base.jsonl 2> $O/scenes_base.err; cat $O/scenes_base.jsonl
cd /tmp; export TMPDIR=/tmp
for c in config3_s6 config4_dense config5_dense256; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$c -o t -- python $GRAFT_REPO_ROOT/tools/bench_scenes.py $c > $O/trace_$c.jsonl 2> $O/trace_$c.err; echo "trace $c rc=$?"
  f=$(find $O/trace_$c -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${c}_kernel_stats.csv
  find $O/trace_$c -name '*kernel_trace.csv' -delete; find $O/trace_$c -name '*.db' -delete
  tail -1 $O/trace_$c.jsonl
done
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k config3 > $O/pytest_config3.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_config3.log
