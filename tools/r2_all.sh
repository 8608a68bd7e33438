# the whole measurement pass in one gpurun call: counters and traces (tools/r2_measure.sh), profiles/traffic.json from them
# (tools/make_traffic.py, stamped with the build id), then the judged line (tools/r2_bench.sh).  traffic.json comes back as
# gpurun_out/$TAG/traffic.json: copy it to profiles/ on the build host.
TAG=${1:-r2_final}; LABEL=${2:-$TAG}
cd $GRAFT_REPO_ROOT
bash tools/r2_measure.sh $TAG || exit 1
ID=$(cat gpurun_out/$TAG/build_id.txt)
python tools/make_traffic.py $ID "$LABEL" config2_s1_1080p=gpurun_out/${TAG}_pmc_config2_s1 config5_dense256_4k=gpurun_out/${TAG}_pmc_config5_dense256 \
    config4_dense_4k=gpurun_out/${TAG}_pmc_config4_dense config3_s6_sky_clouds_restir_1080p=gpurun_out/${TAG}_pmc_config3_s6 > /dev/null || exit 1
cp profiles/traffic.json gpurun_out/$TAG/traffic.json
bash tools/r2_bench.sh $TAG
