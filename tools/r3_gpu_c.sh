# round 3: why the carried schedule is slow -- poll policies and the finish-all diagnostic
TAG=${1:-r3c}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" VRT_BENCH_STEPS=30 timeout -k 10 200 python tools/bench_scenes.py config2_s1 shard_1of8 2>> $O/ab.err | grep -o '"name.*"temporal_ms": [0-9.]*'; }
run VRT_PIPE=overlap
run VRT_PIPE=carry
run VRT_PIPE=carry VRT_LIB_PATH=build_variants/libvrt_poll0.so
run VRT_PIPE=carry VRT_LIB_PATH=build_variants/libvrt_poll1.so
run VRT_PIPE=carry VRT_CARRY_FINISH_ALL=1
run VRT_OVERLAP=0
