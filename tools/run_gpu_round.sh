cd $GRAFT_REPO_ROOT
O=gpurun_out/r41; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_sharded.py -x -q 2>&1 | tail -30
EM2_BENCH_SHARE_DEVICE=1 EM2_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 EM2_BLOCKS_PER_CU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 2 --steps 2 --warmup 1 --cells 200000 --genes 3000 --no-cpu-baseline --check-rows 96 > $O/two.json 2> $O/two.err; tail -c 3000 $O/two.json; tail -5 $O/two.err
