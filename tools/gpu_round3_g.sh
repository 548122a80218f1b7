#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
timeout 600 python bench.py --no-cpu --config c4 --steps 3 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err; tail -1 $O/bench_c4.err
timeout 600 python bench.py --no-cpu --config c2 --steps 10 --warmup 2 > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.err
timeout 600 python bench.py --no-cpu --steps 3 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err; tail -1 $O/bench_c3.err
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import json; d=json.load(open('$O/bench_c4.json')); print(d['value'], d['roofline']['frac_hw_both'])"
