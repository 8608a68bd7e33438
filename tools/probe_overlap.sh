cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2_probe; mkdir -p $O
echo "== plain, gate on"; timeout -k 5 60 python $GRAFT_REPO_ROOT/tools/probe_overlap.py > $O/plain.log 2>&1; echo rc=$?; tail -2 $O/plain.log
echo "== pmc, overlap on, gate OFF"; VRT_DRAIN_GATE=0 timeout -k 5 90 rocprofv3 --kernel-trace --pmc SQ_WAVES --output-format csv -d $O/pmc_nogate -o x -- python $GRAFT_REPO_ROOT/tools/probe_overlap.py > $O/pmc_nogate.log 2>&1; echo rc=$?; tail -3 $O/pmc_nogate.log
echo "== pmc, overlap on, gate ON"; timeout -k 5 90 rocprofv3 --kernel-trace --pmc SQ_WAVES --output-format csv -d $O/pmc_gate -o x -- python $GRAFT_REPO_ROOT/tools/probe_overlap.py > $O/pmc_gate.log 2>&1; echo rc=$?; tail -3 $O/pmc_gate.log
echo "== after: plain again"; timeout -k 5 60 python $GRAFT_REPO_ROOT/tools/probe_overlap.py 2 2>&1 | tail -1
