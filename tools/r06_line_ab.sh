# round 6: theta resampling, line engine against the stage chains, through the stage lab (pxs_debug_chain) and the C4 / C2 bench
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r06_line_ab}; mkdir -p $O
for v in 1 0; do PXS_THETA_LINE=$v python tools/chain_lab.py c4 5 2>&1 | tail -1 | sed "s/^/line=$v /"; done | tee $O/chain_lab_c4.txt
if [ "$2" = "bench" ]; then
for v in 1 0; do
PXS_THETA_LINE=$v timeout 600 python bench.py --config c4 --no-cpu > $O/c4_line$v.json 2> $O/c4_line$v.err; python - <<PY
import json; d=json.load(open("$O/c4_line$v.json")); print("c4 line=$v", d["ms_per_step"], d.get("stage_ms_per_step"), d.get("roundtrip_rms_error"))
PY
done
fi
