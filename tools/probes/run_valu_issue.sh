cd /tmp; export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/tools/probes/valu_issue; O=$GRAFT_REPO_ROOT/gpurun_out/r2_valu; mkdir -p $O
timeout -k 5 120 $P | tee $O/valu_issue.txt
timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc1 -o v -- $P > $O/pmc1.log 2>&1; echo rc=$?
timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/pmc2 -o v -- $P > $O/pmc2.log 2>&1; echo rc=$?
rocprofv3 -L 2>/dev/null | grep -i "SQ_INSTS_VALU\|SQ_ACTIVE_INST\|VALU" | head -40 > $O/counters_valu.txt
