# round 3, the measurement pass, part B (after profiles/traffic.json of part A is in place): scene table incl. the shards the
# scaling prediction uses, present rates, kernel traces of the bench command (config 2) and of configs 3 / 4 / 5, the default
# schedule under --pmc, then the judged line (bench.py).  Everything lands in gpurun_out/$TAG.
TAG=${1:-r3_final}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/bench_scenes.py > $O/scenes.jsonl 2> $O/scenes.err; echo "scenes: $(grep -c name $O/scenes.jsonl)"
for mode in 1 async async8; do for lag in 1 2; do
  [ $mode == 1 ] && [ $lag == 2 ] && continue
  VRT_BENCH_STEPS=40 VRT_BENCH_FETCH_EACH=$mode VRT_BENCH_FETCH_LAG=$lag python tools/bench_scenes.py config2_s1 config4_dense 2>/dev/null | sed "s/_d8\"/_d8_present_${mode}_lag${lag}\"/; s/_1gpu\"/_1gpu_present_${mode}_lag${lag}\"/" >> $O/present.jsonl
done; done; echo "present: $(grep -c name $O/present.jsonl)"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> $O/trace.err; echo "trace rc=$?"
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
for c in config3_s6 config4_dense config5_dense256; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$c -o t -- python $GRAFT_REPO_ROOT/tools/bench_scenes.py $c > $O/trace_$c.jsonl 2> $O/trace_$c.err; echo "trace $c rc=$?"
  f=$(find $O/trace_$c -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${c}_kernel_stats.csv
done
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU --output-format csv -d $O/pmc_default -o p -- python $GRAFT_REPO_ROOT/tools/probe_overlap.py > $O/pmc_default_schedule.txt 2>&1; echo "pmc default schedule rc=$?"
rm -rf $O/pmc_default
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'traffic', r['traffic'], r.get('all_launches_in_flight',{}).get('frac'))
[print(s.get('name'), s.get('value'), s.get('ms_per_step'), s.get('roofline',{}).get('frac'), s.get('roofline',{}).get('traffic'), s.get('error')) for s in d['secondary']]
print(d['cpu_baseline'])"
