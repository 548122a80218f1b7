#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
for cfg in c3 c4; do
  python tools/chain_lab.py $cfg 3 2>> $O/lab.err | tee -a $O/lab.jsonl
  PXS_CH_BLOCKED=0 python tools/chain_lab.py $cfg 3 2>> $O/lab.err | tee -a $O/lab.jsonl
done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --no-cpu --steps 3 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err; tail -1 $O/bench_c3.err
timeout 300 python tools/adj_bench.py > $O/adj_bench.log 2>&1; tail -5 $O/adj_bench.log
