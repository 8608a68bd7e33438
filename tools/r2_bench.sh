# the judged line, after profiles/traffic.json has been regenerated for this build
TAG=${1:-r2_final}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'traffic', r['traffic'], r.get('valu_issue'))
[print(s.get('name'), s.get('value'), s.get('ms_per_step'), s.get('roofline',{}).get('frac'), s.get('roofline',{}).get('traffic'), s.get('error')) for s in d['secondary']]"
