cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for L in 1024 2048; do
EM2_BENCH_SHARE_DEVICE=1 EM2_BENCH_BACKEND=gloo EM2_BLOCKS_PER_CU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --cells 400000 --genes 10000 --lsh-count $L --no-cpu-baseline > gpurun_out/bench_2ranks_$L.json 2> gpurun_out/bench_2ranks_$L.err
tail -3 gpurun_out/bench_2ranks_$L.err | cut -c1-300
python3 -c "
import json
d=json.loads(open('gpurun_out/bench_2ranks_$L.json').read().strip().splitlines()[-1])
print('L=$L', d['n_gpus'], d['ms_per_step'], d.get('parity_check'), d.get('stages_ms_max_over_ranks'), (d.get('row_shard_leg') or {}).get('ms_per_step'), d['config'].get('scan'))"
done
