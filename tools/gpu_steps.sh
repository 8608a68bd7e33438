#!/bin/bash
# One parametrised script for the GPU box (replaces the per-run ab_*.sh / r2_*.sh / r3_*.sh scripts of earlier rounds):
#     gpurun --timeout 900 -- 'bash tools/gpu_steps.sh TAG step [step ...]'
# Every step writes under gpurun_out/TAG and prints one status line; a failed or timed-out GPU step ends the script
# (no further GPU step is started after one that was killed).  Steps:
#   smoke           __graft_entry__.smoke()
#   tests           the whole GPU suite;           tests:EXPR  -> pytest -k EXPR
#   bench           the judged line (bench.py) + a short digest
#   scenes[:names]  tools/bench_scenes.py [names, comma separated]
#   refidx          tools/ref_indexing_diff.py on configs 4 and 5 (GPU)
#   trace:CASE      rocprofv3 --kernel-trace --stats of bench_scenes.py CASE (CASE = bench: the bench command, config 2)
#   pmc:CASE        tools/pmc.sh CASE + summary
#   diag:NAME:ARGS       tools/diag_regions.py ARGS (comma separated) with build_variants/libvrt_NAME.so (a -DVRT_DIAG_REGIONS build)
#   variant:NAME:CASES   tools/bench_scenes.py CASES (comma separated) with build_variants/libvrt_NAME.so (built beforehand: tools/build_variant.sh)
set -o pipefail
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
fail() { echo "$1 FAILED rc=$2"; exit $2; }
for step in "$@"; do
  name=${step%%:*}; arg=${step#*:}; [ "$arg" == "$step" ] && arg=""
  case $name in
    smoke)
      timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || fail smoke $?
      tail -1 $O/smoke.log ;;
    tests)
      if [ -n "$arg" ]; then timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu -k "$arg" > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; fail tests $?; }
      else timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; fail tests $?; }; fi
      tail -1 $O/pytest_gpu.log ;;
    bench)
      timeout -k 10 600 python bench.py $arg > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; fail bench $?; }
      python tools/bench_digest.py $O/bench.json ;;
    scenes)
      timeout -k 10 900 python tools/bench_scenes.py ${arg//,/ } > $O/scenes${arg:+_}${arg//,/_}.jsonl 2> $O/scenes.err || { tail -5 $O/scenes.err; fail scenes $?; }
      python -c "
import json,sys
for l in open('$O/scenes${arg:+_}${arg//,/_}.jsonl'):
    d=json.loads(l); print(' ', d.get('name'), d.get('mpaths_per_s'), {k:v for k,v in d.items() if k.endswith('_ms')})" ;;
    refidx)
      for c in 4 5; do timeout -k 10 600 python tools/ref_indexing_diff.py --backend gpu --config $c > $O/refidx_config$c.json 2> $O/refidx.err || { tail -5 $O/refidx.err; fail refidx $?; }; cat $O/refidx_config$c.json; done ;;
    trace)
      ( cd /tmp; export TMPDIR=/tmp
        if [ "$arg" == "bench" ]; then timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> $O/trace_bench.err
        else timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$arg -o t -- python $GRAFT_REPO_ROOT/tools/bench_scenes.py $arg > $O/trace_$arg.jsonl 2> $O/trace_$arg.err; fi ) || fail trace:$arg $?
      f=$(find $O/trace_$arg -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${arg}_kernel_stats.csv && head -8 $O/${arg}_kernel_stats.csv | cut -c1-160
      find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete ;;
    pmc)
      bash tools/pmc.sh $arg ${TAG}_pmc_$arg > /dev/null 2>&1 || fail pmc:$arg $?
      python tools/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$arg > $O/pmc_$arg.txt; rm -rf gpurun_out/${TAG}_pmc_$arg
      echo "pmc $arg: $(grep -c mean $O/pmc_$arg.txt) rows" ;;
    variant)
      IFS=: read -r vname vcases <<< "$arg"
      VRT_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/libvrt_$vname.so timeout -k 10 600 python tools/bench_scenes.py ${vcases//,/ } > $O/variant_$vname.jsonl 2> $O/variant_$vname.err || { tail -5 $O/variant_$vname.err; fail variant:$vname $?; }
      python -c "
import json
for l in open('$O/variant_$vname.jsonl'):
    d=json.loads(l); print('  [$vname]', d.get('name'), d.get('mpaths_per_s'), {k:v for k,v in d.items() if k.endswith('_ms')})" ;;
    diag)   # diag:NAME:ARGS -> tools/diag_regions.py ARGS with build_variants/libvrt_NAME.so (a -DVRT_DIAG_REGIONS build)
      IFS=: read -r vname dargs <<< "$arg"
      VRT_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/libvrt_$vname.so timeout -k 10 600 python tools/diag_regions.py ${dargs//,/ } > $O/diag_$vname.txt 2> $O/diag_$vname.err || { tail -5 $O/diag_$vname.err; fail diag:$vname $?; }
      cat $O/diag_$vname.txt ;;
    *) echo "unknown step $step"; exit 64 ;;
  esac
done
