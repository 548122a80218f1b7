O=gpurun_out/s3a; mkdir -p $O
timeout 600 python tools/finecc_check.py big > $O/finecc_check.txt 2>&1; tail -12 $O/finecc_check.txt
timeout 900 python -m pytest tests/test_sht_parity.py tests/test_baseline_configs.py -m gpu -x -q > $O/pytest_part.log 2>&1; tail -3 $O/pytest_part.log
for v in interpolant finecc; do PXS_ANALYSIS=$v python tools/chain_lab.py c3 3 >> $O/chain_lab.jsonl 2>> $O/lab.err; done
PXS_THETA_DUCC_NCC=0 python tools/chain_lab.py c3 3 >> $O/chain_lab.jsonl 2>> $O/lab.err
PXS_ANALYSIS=finecc python tools/chain_lab.py c4 3 >> $O/chain_lab.jsonl 2>> $O/lab.err
PXS_THETA_DUCC_NCC=0 python tools/chain_lab.py c4 3 >> $O/chain_lab.jsonl 2>> $O/lab.err
cat $O/chain_lab.jsonl
timeout 600 python bench.py --no-cpu > $O/bench_c3_default.json 2> $O/bench_c3_default.err; tail -1 $O/bench_c3_default.err
PIXELL_AMD_ANALYSIS=finecc PXS_BENCH_NO_WEIGHTS=1 timeout 600 python bench.py --no-cpu > $O/bench_c3_finecc.json 2> $O/bench_c3_finecc.err; tail -1 $O/bench_c3_finecc.err
PXS_THETA_DUCC_NCC=0 PXS_BENCH_NO_WEIGHTS=1 timeout 600 python bench.py --no-cpu > $O/bench_c3_oldncc.json 2> $O/bench_c3_oldncc.err; tail -1 $O/bench_c3_oldncc.err
python - <<'PY'
import json
for n in ["default","finecc","oldncc"]:
    try:
        d=json.loads(open("gpurun_out/s3a/bench_c3_%s.json"%n).read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["stage_ms_per_step"], d.get("analysis_weights",{}).get("ms_per_step"), d["roundtrip_rms_error"])
    except Exception as e: print(n, "failed", e)
PY
