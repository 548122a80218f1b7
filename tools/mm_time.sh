#!/bin/bash
# phase timers of leg_ana_s0_mm (lab build -DPXS_LAB_MMTIME, tools/libpxsht_mmtime.so) + the plain bench; usage: mm_time.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=${1:-a}; O=$R/gpurun_out/mmtime_$TAG; mkdir -p $O; cd $R
PIXELL_AMD_LIB=$R/tools/libpxsht_mmtime.so PXS_BENCH_NBATCH=16 timeout 300 python bench.py --config c4 --no-cpu --steps 1 --warmup 0 2>&1 | grep -E "mm_prof|stage ms" | tail -3 | tee $O/mm_prof.txt
timeout 300 python bench.py --config c4 --no-cpu --steps 3 > $O/c4.json 2> $O/c4.err; grep -E "stage ms|rms error" $O/c4.err | cut -c1-200
timeout 600 python -m pytest tests -m gpu -x -q -k "mm_analysis or batched or config4" 2>&1 | tail -3
