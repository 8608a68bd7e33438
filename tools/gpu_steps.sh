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
#   timeline:CASE[:MODE[:LAG]]  launches and copies in time (tools/timeline.py)
#   predict:CONFIG[:N,..]  predicted N-GPU step times (tools/predict_scaling.py)
#   traffic         counters of every config -> profiles/traffic.json for this build;  present: present rates;  pmcdefault: the default schedule under --pmc
#   diag:NAME:ARGS       tools/diag_regions.py ARGS (comma separated) with build_variants/libvrt_NAME.so (a -DVRT_DIAG_REGIONS build)
#   variant:NAME:CASES   tools/bench_scenes.py CASES (comma separated) with build_variants/libvrt_NAME.so (built beforehand: tools/build_variant.sh)
set -o pipefail
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
fail() { local rc=${2:-1}; [ "$rc" == 0 ] && rc=1; echo "$1 FAILED rc=$rc"; exit $rc; }   # (never 0: a later step must not start)
for step in "$@"; do
  name=${step%%:*}; arg=${step#*:}; [ "$arg" == "$step" ] && arg=""
  case $name in
    smoke)
      timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || fail smoke $?
      tail -1 $O/smoke.log ;;
    tests)
      if [ -n "$arg" ]; then timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu -k "$arg" > $O/pytest_gpu.log 2>&1 || { rc=$?; tail -30 $O/pytest_gpu.log; fail tests $rc; }
      else timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { rc=$?; tail -30 $O/pytest_gpu.log; fail tests $rc; }; fi
      tail -1 $O/pytest_gpu.log ;;
    bench)
      timeout -k 10 600 python bench.py $arg > $O/bench.json 2> $O/bench.err || { rc=$?; tail -5 $O/bench.err; fail bench $rc; }
      python tools/bench_digest.py $O/bench.json ;;
    scenes)
      timeout -k 10 900 python tools/bench_scenes.py ${arg//,/ } > $O/scenes${arg:+_}${arg//,/_}.jsonl 2> $O/scenes.err || { rc=$?; tail -5 $O/scenes.err; fail scenes $rc; }
      python -c "
import json,sys
for l in open('$O/scenes${arg:+_}${arg//,/_}.jsonl'):
    d=json.loads(l); print(' ', d.get('name'), d.get('mpaths_per_s'), {k:v for k,v in d.items() if k.endswith('_ms')})" ;;
    refidx)
      for c in 4 5; do timeout -k 10 600 python tools/ref_indexing_diff.py --backend gpu --config $c > $O/refidx_config$c.json 2> $O/refidx.err || { rc=$?; tail -5 $O/refidx.err; fail refidx $rc; }; cat $O/refidx_config$c.json; done ;;
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
      VRT_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/libvrt_$vname.so timeout -k 10 600 python tools/bench_scenes.py ${vcases//,/ } > $O/variant_$vname.jsonl 2> $O/variant_$vname.err || { rc=$?; tail -5 $O/variant_$vname.err; fail variant:$vname $rc; }
      python -c "
import json
for l in open('$O/variant_$vname.jsonl'):
    d=json.loads(l); print('  [$vname]', d.get('name'), d.get('mpaths_per_s'), {k:v for k,v in d.items() if k.endswith('_ms')})" ;;
    traffic)   # counters per config (tools/pmc.sh) -> profiles/traffic.json stamped with the build id; comes back as gpurun_out/TAG/traffic.json
      python -c "from voxel_rt2_amd import _lib; print(_lib.build_id())" > $O/build_id.txt 2>/dev/null
      ID=$(cat $O/build_id.txt); echo "build $ID"
      for c in config2_s1 config5_dense256 config4_dense config3_s6; do
        bash tools/pmc.sh $c ${TAG}_pmc_$c > /dev/null 2>&1 || fail traffic:$c $?
        python tools/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$c > $O/pmc_$c.txt
        echo "pmc $c: $(grep -c mean $O/pmc_$c.txt) rows"
      done
      python tools/make_traffic.py $ID "$TAG" config2_s1_1080p=gpurun_out/${TAG}_pmc_config2_s1 config5_dense256_4k=gpurun_out/${TAG}_pmc_config5_dense256 \
          config4_dense_4k=gpurun_out/${TAG}_pmc_config4_dense config3_s6_sky_clouds_restir_1080p=gpurun_out/${TAG}_pmc_config3_s6 > $O/traffic_digest.json || fail traffic 1
      cp profiles/traffic.json $O/traffic.json
      rm -rf gpurun_out/${TAG}_pmc_* ;;
    present)   # the frame copied to the host after every step: blocking, asynchronous f32, asynchronous 8 bit
      for mode in 1 async async8; do for lag in 1 2 3; do
        [ $mode == 1 ] && [ $lag != 1 ] && continue
        [ $mode == async ] && [ $lag == 3 ] && continue
        VRT_BENCH_STEPS=120 VRT_BENCH_FETCH_EACH=$mode VRT_BENCH_FETCH_LAG=$lag timeout -k 10 300 python tools/bench_scenes.py config2_s1 config4_dense 2>/dev/null | sed "s/_d8\"/_d8_present_${mode}_lag${lag}\"/; s/_1gpu\"/_1gpu_present_${mode}_lag${lag}\"/" >> $O/present.jsonl
      done; done; echo "present: $(grep -c name $O/present.jsonl)" ;;
    timeline)   # timeline:CASE[:MODE[:LAG]] -> kernel + copy trace of tools/bench_scenes.py CASE (MODE: VRT_BENCH_FETCH_EACH), digested by tools/timeline.py
      IFS=: read -r tcase tmode tlag <<< "$arg"
      ( cd /tmp; export TMPDIR=/tmp VRT_BENCH_STEPS=60 VRT_BENCH_FETCH_EACH=$tmode VRT_BENCH_FETCH_LAG=${tlag:-1}
        timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl_$tcase$tmode -o t -- python $GRAFT_REPO_ROOT/tools/bench_scenes.py $tcase > $O/tl_$tcase$tmode.jsonl 2> $O/tl_$tcase$tmode.err ) || fail timeline:$arg $?
      python tools/timeline.py $O/tl_$tcase$tmode > $O/timeline_$tcase${tmode:+_}$tmode${tlag:+_lag}$tlag.txt; rm -rf $O/tl_$tcase$tmode
      head -12 $O/timeline_$tcase${tmode:+_}$tmode${tlag:+_lag}$tlag.txt ;;
    predict)   # predict:CONFIG[:N,N,...] -> tools/predict_scaling.py (bench.py --gpus N replayed rank by rank on this GPU)
      IFS=: read -r pcfg pn <<< "$arg"
      timeout -k 10 600 python tools/predict_scaling.py $pcfg ${pn//,/ } > $O/predict_$pcfg.jsonl 2> $O/predict_$pcfg.err || { rc=$?; tail -5 $O/predict_$pcfg.err; fail predict:$arg $rc; }
      python -c "
import json
for l in open('$O/predict_$pcfg.jsonl'):
    d=json.loads(l)
    if 'n_gpus' in d: print('  $pcfg N =', d['n_gpus'], 'equal', d['equal_speedup'], 'balanced', d['balanced_speedup'], {k: v['speedup'] for k, v in d.items() if k.startswith('stripes')})" ;;
    pmcdefault)
      ( cd /tmp; export TMPDIR=/tmp; timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU --output-format csv -d $O/pmc_default -o p -- python $GRAFT_REPO_ROOT/tools/probe_overlap.py > $O/pmc_default_schedule.txt 2>&1 ) || fail pmcdefault $?
      rm -rf $O/pmc_default; tail -2 $O/pmc_default_schedule.txt ;;
    diag)   # diag:NAME:ARGS -> tools/diag_regions.py ARGS with build_variants/libvrt_NAME.so (a -DVRT_DIAG_REGIONS build)
      IFS=: read -r vname dargs <<< "$arg"
      VRT_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/libvrt_$vname.so timeout -k 10 600 python tools/diag_regions.py ${dargs//,/ } > $O/diag_$vname.txt 2> $O/diag_$vname.err || { rc=$?; tail -5 $O/diag_$vname.err; fail diag:$vname $rc; }
      cat $O/diag_$vname.txt ;;
    *) echo "unknown step $step"; exit 64 ;;
  esac
done
