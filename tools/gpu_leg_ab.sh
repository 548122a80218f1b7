#!/bin/bash
# same-box A/B of the Legendre kernels through bench.py (--no-cpu): stock library, then every library named on the command line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${LAB_OUT:-legab}; mkdir -p $O; CFG=${LAB_CFG:-c3}
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -6 | tee $O/smi.txt
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py --config $CFG --no-cpu --steps ${LAB_STEPS:-4} > $O/$tag.json 2> $O/$tag.err
  python - "$O/$tag.json" "$tag" <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1])); print(sys.argv[2], "ms/step", r["ms_per_step"], r["stage_ms_per_step"])
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
run stock PXS_DUMMY=1
for l in "$@"; do run $(basename $l .so) PIXELL_AMD_LIB=$PWD/$l; done
for e in $LAB_ENVS; do run "$(echo $e | tr '=,' '__')" $(echo $e | tr ',' ' '); done
