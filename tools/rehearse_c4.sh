# Rehearsal of the multi-rank bench paths on a ONE-GPU box: two ranks share the device, the alm gather goes through gloo (host
# memory).  Proves the sharding / gather / timing logic of `bench.py --gpus N` end to end; the RCCL runs are the driver's.
O=gpurun_out/rehearsal; mkdir -p $O
export PXS_BENCH_BACKEND=gloo PXS_BENCH_NBATCH=8
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --config c4 --steps 2 --warmup 1 > $O/c4_2ranks_gloo.json 2> $O/c4_2ranks_gloo.err; tail -3 $O/c4_2ranks_gloo.err; cut -c1-600 $O/c4_2ranks_gloo.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 2 --config c2 --steps 2 --warmup 1 > $O/c2_2ranks_gloo.json 2> $O/c2_2ranks_gloo.err; tail -2 $O/c2_2ranks_gloo.err; cut -c1-400 $O/c2_2ranks_gloo.json
