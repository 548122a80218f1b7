O=gpurun_out/s3b; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --no-cpu > $O/bench_c3.json 2> $O/bench_c3.err; tail -1 $O/bench_c3.err
timeout 600 python bench.py --no-cpu --config c4 > $O/bench_c4.json 2> $O/bench_c4.err; tail -1 $O/bench_c4.err
timeout 600 python bench.py --no-cpu --config c2 --steps 10 --warmup 2 > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.err
