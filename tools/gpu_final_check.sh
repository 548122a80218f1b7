# what the driver runs at round end, plus the 2-D FFT A/B: GPU tests, smoke, the default bench line
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-final}; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
t0=$(date +%s); timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench wall time $(( $(date +%s) - t0 )) s"; tail -1 $O/bench_default.err; tail -c 300 $O/bench_default.json
for i in 1 2; do python tools/fft2_bench.py 2>/dev/null | tee -a $O/fft2.txt; if [ -f tools/libpxsht_nor7.so ]; then PIXELL_AMD_LIB=$PWD/tools/libpxsht_nor7.so python tools/fft2_bench.py 2>/dev/null | tee -a $O/fft2.txt; fi; done
exit 0
