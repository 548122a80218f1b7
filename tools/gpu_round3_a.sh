#!/bin/bash
# first GPU pass of round 3: GPU tests, bench lines c3 / c4 / c5 / c2 (no CPU legs)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --no-cpu --steps 5 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err; tail -2 $O/bench_c3.err
timeout 600 python bench.py --no-cpu --config c4 --steps 3 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err; tail -2 $O/bench_c4.err
timeout 600 python bench.py --no-cpu --config c2 --steps 10 --warmup 2 > $O/bench_c2.json 2> $O/bench_c2.err; tail -2 $O/bench_c2.err
PXS_BENCH_NREAL=10 timeout 600 python bench.py --no-cpu --config c5 --steps 2 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err; tail -3 $O/bench_c5.err
cat $O/bench_c3.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_hw_both'))"
cat $O/bench_c4.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['stage_ms_per_step'])"
cat $O/bench_c5.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_realisation'], d['value'], d['stage_ms_per_realisation'])"
