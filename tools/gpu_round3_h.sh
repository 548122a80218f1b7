#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
for cfg in c3 c4 c5 c2; do PXS_CHAIN_VERBOSE=1 python tools/chain_lab.py $cfg 3 2>> $O/lab.err | tee -a $O/lab.jsonl; done
grep "ring chain" $O/lab.err | sort -u
timeout 900 python -m pytest tests/test_sht_parity.py tests/test_fft_parity.py tests/test_baseline_configs.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 600 python bench.py --no-cpu --steps 3 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err; tail -1 $O/bench_c3.err
timeout 600 python bench.py --no-cpu --config c4 --steps 3 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err; tail -1 $O/bench_c4.err
