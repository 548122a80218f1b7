# round 6: the single-kernel theta engine on the GPU -- parity tests that go through to_cc / from_cc_adjoint, then C4 / C2 with and without it
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r06_line}; mkdir -p $O
PXS_CHAIN_VERBOSE=1 timeout 900 python -m pytest tests/test_sht_parity.py -x -q -m gpu -k "ducc0_route or adjoint_analysis or weights_analysis or grid" > $O/pytest_line.log 2>&1; tail -3 $O/pytest_line.log; grep "theta line" $O/pytest_line.log | sort | uniq | head -20
for v in 1 0; do
PXS_THETA_LINE=$v timeout 600 python bench.py --config c4 --no-cpu > $O/c4_line$v.json 2> $O/c4_line$v.err; python - <<PY
import json; d=json.load(open("$O/c4_line$v.json")); print("c4 line=$v", d["ms_per_step"], d.get("stage_ms_per_step"), d.get("roundtrip_rms_error"))
PY
done
for v in 1 0; do
PXS_THETA_LINE=$v timeout 600 python bench.py --config c2 --no-cpu > $O/c2_line$v.json 2> $O/c2_line$v.err; python - <<PY
import json; d=json.load(open("$O/c2_line$v.json")); print("c2 line=$v", d["ms_per_step"], d.get("stage_ms_per_step"), d.get("roundtrip_rms_error"))
PY
done
