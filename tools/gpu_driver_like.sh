# what the driver runs at round end, in its own form: GPU tests, smoke, `bench.py --gpus 1 --steps 20 --warmup 5`
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-driverlike}; mkdir -p $O
PXS_REQUIRE_FULL=1 timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu exit code $?"; tail -2 $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke exit code $?"; tail -1 $O/smoke.log
t0=$(date +%s); timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench exit code $?"; echo "bench --steps 20 --warmup 5 wall time $(( $(date +%s) - t0 )) s"; tail -1 $O/bench_20.err; tail -c 300 $O/bench_20.json
