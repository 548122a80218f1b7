#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
timeout 900 python -m pytest tests/test_fft_parity.py tests/test_flatsky.py tests/test_uharm.py -m gpu -x -q > $O/pytest_fft.log 2>&1; tail -3 $O/pytest_fft.log
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err; tail -2 $O/bench_c3.err
python -c "import json; d=json.load(open('$O/bench_c3.json')); print(d['fft']); print(d['cpu_baseline'])"
PXS_BENCH_NREAL=10 timeout 600 python bench.py --no-cpu --config c5 --steps 2 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err; tail -1 $O/bench_c5.err
python -c "import json; d=json.load(open('$O/bench_c5.json')); print(d['ms_per_realisation'], d['stage_ms_per_realisation'], d['fft'])"
