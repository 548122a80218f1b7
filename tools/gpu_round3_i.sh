#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
PXS_BENCH_FORCE_PG=1 timeout 600 python bench.py --no-cpu --config c2 --steps 3 --warmup 1 > $O/bench_c2_pg.json 2> $O/bench_c2_pg.err; tail -3 $O/bench_c2_pg.err; python -c "import json; d=json.load(open('$O/bench_c2_pg.json')); print(d['ms_per_step'], d.get('rccl_ranks_seen'), d.get('collective'))"
PXS_BENCH_FORCE_PG=1 PXS_BENCH_NBATCH=8 timeout 600 python bench.py --no-cpu --config c4 --steps 2 --warmup 1 > $O/bench_c4_pg.json 2> $O/bench_c4_pg.err; tail -2 $O/bench_c4_pg.err; python -c "import json; d=json.load(open('$O/bench_c4_pg.json')); print(d['ms_per_step'], d.get('rccl_ranks_seen'), d.get('collective'))"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu --config c2 --steps 2 --warmup 1 > $O/bench_c2_torchrun.json 2> $O/bench_c2_torchrun.err; tail -1 $O/bench_c2_torchrun.err; tail -c 300 $O/bench_c2_torchrun.json
