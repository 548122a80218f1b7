# round 6: the GPU test suite + the bench lines of the round.  usage (GPU box): bash tools/r06_gpu_suite.sh <tag> [quick]
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r06_suite}; mkdir -p $O
if [ "$2" = "quick" ]; then timeout 1500 python -m pytest tests/test_theta_line.py tests/test_baseline_configs.py -x -q -m gpu > $O/pytest_gpu.log 2>&1
else PXS_REQUIRE_FULL=1 timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; fi
tail -3 $O/pytest_gpu.log
for c in c4 c2; do
timeout 600 python bench.py --config $c --no-cpu > $O/bench_$c.json 2> $O/bench_$c.err; python - <<PY
import json; d=json.load(open("$O/bench_$c.json")); print("$c", d["ms_per_step"], d["value"], d.get("stage_ms_per_step"), d.get("roundtrip_rms_error"))
PY
done
