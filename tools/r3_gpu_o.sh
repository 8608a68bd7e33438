# randomised parity sweeps on the shipped build: scenario scripts (every buffer after every accumulate) and sizes / depths / shards
TAG=${1:-r3o}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "from voxel_rt2_amd import _lib; print(_lib.build_id())" > $O/build_id.txt 2>/dev/null; echo "build $(cat $O/build_id.txt)"
timeout -k 10 420 python tools/soak_scenarios.py --gpu 140 31 > $O/soak_scenarios_a.txt 2>&1; echo "scenarios a rc=$?"; tail -2 $O/soak_scenarios_a.txt
SOAK_SCALE=4 timeout -k 10 300 python tools/soak_scenarios.py --gpu 30 32 > $O/soak_scenarios_b.txt 2>&1; echo "scenarios b (x4 frames) rc=$?"; tail -2 $O/soak_scenarios_b.txt
timeout -k 10 300 python tools/soak_parity.py 60 33 > $O/soak_parity.txt 2>&1; echo "parity rc=$?"; tail -2 $O/soak_parity.txt
