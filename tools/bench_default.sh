#!/bin/bash
# the driver's command (python bench.py, defaults) with its wall time; prints the headline and the `configs` block in short
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05b}; mkdir -p $O; cd $R; shift
t0=$(date +%s.%N); python bench.py "$@" > $O/bench_default.json 2> $O/bench_default.err; rc=$?; t1=$(date +%s.%N)
python -c "print(\"rc $rc wall %.1f s\" % ($t1 - $t0))"
grep -E "stage ms|failed|Error|error|Traceback" $O/bench_default.err | cut -c1-220 | tail -20
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"], "bench_wall_s", d.get("bench_wall_s"), "roofline", d["roofline"]["frac"], d["roofline"].get("frac_hw"))
for k,v in d.get("configs",{}).items(): print(k, {kk:v.get(kk) for kk in ("ms_per_step","value","unit","ms_per_realisation","leg_wall_s","roundtrip_rms_error","error") if v.get(kk) is not None}, "frac", v.get("roofline",{}).get("frac"), v.get("roofline",{}).get("frac_hw_both"))
PY
