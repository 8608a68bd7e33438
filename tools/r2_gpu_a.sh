# round-2 GPU pass A: full GPU suite, bench line, kernel trace, PMC under the default (overlapped) schedule
O=$GRAFT_REPO_ROOT/gpurun_out/r2a; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['launch_pipeline']); 
[print(s.get('name'), s.get('value'), s.get('ms_per_step'), s.get('roofline',{}).get('frac'), s.get('error')) for s in d['secondary']]; print(d['cpu_baseline']['value'])"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> $O/trace.err; echo "trace rc=$?"
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU --output-format csv -d $O/pmc_overlap -o p -- python $GRAFT_REPO_ROOT/tools/probe_overlap.py > $O/pmc_overlap.log 2>&1; echo "pmc with default schedule rc=$?"; grep -a "^done\|^step 5" $O/pmc_overlap.log
