# rehearsal of bench.py's N > 1 path with more ranks and more steps than tests/test_bench_multi.py (ring of tiles wraps, 3 and 4 ranks)
TAG=${1:-r3l}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
for n in 3 4; do
  VRT_BENCH_REHEARSE=1 MASTER_ADDR=127.0.0.1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 12 --warmup 2 > $O/rehearse_$n.json 2> $O/rehearse_$n.err; echo "rehearsal $n ranks rc=$?"
  grep "rehearsal" $O/rehearse_$n.err
  python -c "
import json; d=json.load(open('$O/rehearse_$n.json')); print(d['n_gpus'], d['value'], d['ms_per_step'], d['config']['sharding'][-60:]); [print(' ', s['name'], s['value'], s['ms_per_step'], s['tile_rows']) for s in d['secondary']]"
done
