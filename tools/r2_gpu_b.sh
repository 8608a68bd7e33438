# round-2 GPU pass B: counter passes per config (isolated launches), FETCH_SIZE calibration
O=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
for c in config2_s1 config5_dense256 config4_dense config3_s6; do bash tools/pmc.sh $c r2b_pmc_$c > /dev/null 2>&1; python tools/pmc_summary.py $O/r2b_pmc_$c > $O/r2b_pmc_$c/summary.txt; echo "$c done: $(grep -c mean $O/r2b_pmc_$c/summary.txt) counter rows"; done
cd /tmp; export TMPDIR=/tmp; mkdir -p $O/r2b_calib
timeout -k 5 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r2b_calib/fetch -o f -- $GRAFT_REPO_ROOT/tools/probes/fetch_calib > $O/r2b_calib/fetch.log 2>&1; echo rc=$?
timeout -k 5 100 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_MISS_sum --output-format csv -d $O/r2b_calib/req -o f -- $GRAFT_REPO_ROOT/tools/probes/fetch_calib > $O/r2b_calib/req.log 2>&1; echo rc=$?
