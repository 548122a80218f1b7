# on the GPU box: GPU tests, then the bench lines of the round (c3 default with all legs, c2, c4 single GPU)
O=gpurun_out/${1:-r2e}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; tail -2 $O/bench_c3.err; cut -c1-1500 $O/bench_c3.json
timeout 600 python bench.py --config c2 --no-cpu > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.err
timeout 900 python bench.py --config c4 --no-cpu > $O/bench_c4.json 2> $O/bench_c4.err; tail -1 $O/bench_c4.err; cut -c1-300 $O/bench_c4.json
