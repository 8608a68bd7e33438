# round 3, the measurement pass on the build that ships, part A: counters per config (tools/pmc.sh) -> profiles/traffic.json stamped
# with the build id (tools/make_traffic.py; it comes back as gpurun_out/$TAG/traffic.json: copy it to profiles/ before part B)
TAG=${1:-r3_final}; LABEL=${2:-$TAG}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -c "from voxel_rt2_amd import _lib; print(_lib.build_id())" > $O/build_id.txt 2>/dev/null
ID=$(cat $O/build_id.txt); echo "build $ID"
for c in config2_s1 config5_dense256 config4_dense config3_s6; do
  bash tools/pmc.sh $c ${TAG}_pmc_$c > /dev/null 2>&1
  python tools/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$c > $O/pmc_$c.txt
  echo "pmc $c: $(grep -c mean $O/pmc_$c.txt) rows"
done
python tools/make_traffic.py $ID "$LABEL" config2_s1_1080p=gpurun_out/${TAG}_pmc_config2_s1 config5_dense256_4k=gpurun_out/${TAG}_pmc_config5_dense256 \
    config4_dense_4k=gpurun_out/${TAG}_pmc_config4_dense config3_s6_sky_clouds_restir_1080p=gpurun_out/${TAG}_pmc_config3_s6 > /dev/null || exit 1
cp profiles/traffic.json $O/traffic.json
rm -rf gpurun_out/${TAG}_pmc_*    # (raw counter csvs: tens of MB)
