cd $GRAFT_REPO_ROOT
O=gpurun_out/r69; mkdir -p $O
EM2_BENCH_SHARE_DEVICE=1 EM2_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 EM2_BLOCKS_PER_CU=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29681 bench.py --gpus 2 --steps 3 --warmup 1 --cells 400000 --genes 10000 --no-cpu-baseline --check-rows 96 > $O/two.json 2> $O/two.err; tail -c 2500 $O/two.json; grep -i "error\|PARITY" $O/two.err | head
